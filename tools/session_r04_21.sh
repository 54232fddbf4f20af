#!/bin/bash
# GPU box, round 4 session 21: <= 4-output-channel transposed conv staged through LDS (convT_small3_kernel) against the
# direct-load kernel at 8 / 4 channels per trip (s3off) and as it was (base): conv parity tests, hot re-issue, step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s21; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_conv.log 2>&1; echo "conv tests rc=$?" > $out/status.txt
tail -5 $out/t_conv.log >> $out/status.txt
cd tools
for v in base s3off s3; do
  MVAE_HIP_LIB=$GRAFT_REPO_ROOT/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 200 python small_conv_probe.py > ../$out/probe_$v.txt 2>&1
done
cd ..
timeout 900 bash tools/ab_matrix.sh 3 \
  "fashionmnist:lib:base,lib:s3off,lib:s3" \
  "celeba:lib:base,lib:s3off,lib:s3" > $out/ab.txt 2>&1
timeout 300 bash tools/ab_matrix.sh 1 "celeba19:lib:base,lib:s3" >> $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt
paste <(cut -c1-57 $out/probe_base.txt) <(cut -c49-57 $out/probe_s3off.txt) <(cut -c49-57 $out/probe_s3.txt)

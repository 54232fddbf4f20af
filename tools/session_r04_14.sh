#!/bin/bash
# GPU box, round 4 session 14: pair-store kernels issue a finished pair's stores during the next item's first k-step (default)
# against at the item boundary (nodefer): conv tests, hot re-issue of the CelebA conv launches, step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s14; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_conv.log 2>&1; echo "conv tests rc=$?" > $out/status.txt
tail -2 $out/t_conv.log >> $out/status.txt
for v in base nodefer; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/conv_$v.txt 2>&1
done
timeout 900 bash tools/ab_matrix.sh 2 \
  "celeba:lib:base,lib:nodefer" \
  "fashionmnist:lib:base,lib:nodefer" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt
paste <(cut -c1-34,46-100 $out/conv_base.txt) <(cut -c46-100 $out/conv_nodefer.txt) | grep -E "enc2|dec2|dec3|op "

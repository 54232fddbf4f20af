#!/bin/bash
# GPU box, round 4 session 19: convT_small2_kernel with U channels fetched per trip (1 / 2 / 4 / 8), with and without
# SLP-packed FMAs: conv parity tests on the default build, hot re-issue of the small-channel conv launches per build,
# step A/B on FashionMNIST / CelebA
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s19; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_conv.log 2>&1; echo "conv tests rc=$?" > $out/status.txt
tail -2 $out/t_conv.log >> $out/status.txt
cd tools
for v in d1 d2 d4 d8 d4noslp d8noslp; do
  MVAE_HIP_LIB=$GRAFT_REPO_ROOT/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 200 python small_conv_probe.py > ../$out/probe_$v.txt 2>&1
done
cd ..
timeout 900 bash tools/ab_matrix.sh 2 \
  "fashionmnist:lib:d1,lib:d4,lib:d8,lib:d8noslp" \
  "celeba:lib:d1,lib:d4,lib:d4noslp" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt
paste <(cut -c1-57 $out/probe_d1.txt) <(cut -c49-57 $out/probe_d2.txt) <(cut -c49-57 $out/probe_d4.txt) <(cut -c49-57 $out/probe_d8.txt) <(cut -c49-57 $out/probe_d4noslp.txt) <(cut -c49-57 $out/probe_d8noslp.txt)

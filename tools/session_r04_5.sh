#!/bin/bash
# GPU box, round 4 session 5: convT_s1 k-depth 16 and a 256-block target for the split conv weight gradients (A/B),
# refreshed per-(call, shape) tables of the two CelebA steps
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s5; rm -rf $out; mkdir -p $out
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_s1bk16.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_s1bk16.log 2>&1; echo "s1bk16 conv tests rc=$?" > $out/status.txt
tail -2 $out/t_s1bk16.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_wgt256.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_wgt256.log 2>&1; echo "wgt256 conv tests rc=$?" >> $out/status.txt
tail -2 $out/t_wgt256.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 2 \
  "celeba19:lib:base,lib:s1bk16,lib:wgt256" \
  "celeba:lib:base,lib:s1bk16,lib:wgt256" \
  "fashionmnist:lib:base,lib:wgt256" > $out/ab.txt 2>&1
timeout 400 bash tools/collect_by_shape.sh "celeba celeba19" > $out/by_shape.log 2>&1
cp gpurun_out/by_shape/r04_* $out/ 2>/dev/null
cat $out/status.txt; cat $out/ab.txt

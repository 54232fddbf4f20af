#!/bin/bash
# GPU box, round 4 session 7: BatchNorm tests after the per-slice form was restricted to where it wins; convT_s1 col2im in
# two 32-column passes (s1half: 17 KB of staging -> six blocks per CU) against four; noslice = no per-slice form at all
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s7; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "batchnorm or conv" > $out/t_default.log 2>&1; echo "default bn+conv tests rc=$?" > $out/status.txt
tail -2 $out/t_default.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_s1half.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_s1half.log 2>&1; echo "s1half conv tests rc=$?" >> $out/status.txt
tail -2 $out/t_s1half.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 \
  "celeba19:lib:base,lib:s1half,lib:oldslice" \
  "celeba:lib:base,lib:s1half" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

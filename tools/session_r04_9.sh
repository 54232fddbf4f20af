#!/bin/bash
# GPU box, round 4 session 9: convT_s1 col2im tap table hoisted out of the per-image loop (default; notaps = per image),
# PoE launches of the bimodal steps cut to one term / expert per block (poe1)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s9; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv or poe" > $out/t_default.log 2>&1; echo "default conv+poe tests rc=$?" > $out/status.txt
tail -2 $out/t_default.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_poe1.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "poe" > $out/t_poe1.log 2>&1; echo "poe1 tests rc=$?" >> $out/status.txt
tail -2 $out/t_poe1.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 \
  "celeba19:lib:base,lib:notaps" \
  "celeba:lib:base,lib:notaps,lib:poe1" \
  "mnist:lib:base,lib:poe1" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

#!/bin/bash
# GPU box, round 4 session 6: convT_s1 with contiguous dy runs (s1co) against the adopted 16-deep k-tile (base) and the old 32 (s1bk32)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s6; rm -rf $out; mkdir -p $out
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_s1co.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_s1co.log 2>&1; echo "s1co conv tests rc=$?" > $out/status.txt
tail -2 $out/t_s1co.log >> $out/status.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_default.log 2>&1; echo "default conv tests rc=$?" >> $out/status.txt
tail -2 $out/t_default.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 2 \
  "celeba19:default,lib:base,lib:s1bk32,lib:s1co" \
  "celeba:default,lib:base,lib:s1bk32,lib:s1co" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

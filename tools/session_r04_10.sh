#!/bin/bash
# GPU box, round 4 session 10: PoE tests on the adopted small-step chunking; terms / experts per block of the many-term
# launch (celeba19): 3 (base) vs 1 (pc1) vs 2 (pc2)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s10; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_reference_names_gpu.py -m gpu -q -k "poe or product or experts or prior" > $out/t_default.log 2>&1; echo "default poe tests rc=$?" > $out/status.txt
tail -2 $out/t_default.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_pc1.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "poe" > $out/t_pc1.log 2>&1; echo "pc1 tests rc=$?" >> $out/status.txt
tail -2 $out/t_pc1.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 "celeba19:lib:base,lib:pc1,lib:pc2" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

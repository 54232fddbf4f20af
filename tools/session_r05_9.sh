#!/bin/bash
# round 5, session 9: re-tune of the multi-item knobs now that the conv kernels hold four blocks per CU (step A/B, x2 interleaved)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s9; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "stats or bce" > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -4 $out/tests.log >> $out/status.txt
V="lib:base,lib:maxk512,lib:minb2048,lib:minb512,lib:items8,lib:items2"
timeout 1500 bash tools/ab_matrix.sh 2 "celeba:$V" "fashionmnist:$V" "celeba19:$V" > $out/ab_retune.txt 2>&1
cat $out/status.txt $out/ab_retune.txt

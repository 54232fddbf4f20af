#!/bin/bash
# round 5, session 8: full GPU suite on the current tree; 128 x 64 vs 64 x 64 tiles for the wide conv-forward-form launches (step A/B)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s8; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -12 $out/tests.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 "celeba:lib:base,lib:nowide" "fashionmnist:lib:base,lib:nowide" "celeba19:lib:base,lib:nowide" > $out/ab_wide.txt 2>&1
cat $out/status.txt $out/ab_wide.txt

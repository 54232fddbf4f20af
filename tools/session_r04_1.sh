#!/bin/bash
# GPU box, round 4 session 1: full GPU suite on the new kernels (XCD-local launch orders, single-launch BatchNorm,
# paired MNIST encoders), the A/B matrix of their switches, PMC traffic of the three launches VERDICT r3 names.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s1; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q --maxfail=15 > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -5 $out/tests.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 2 \
  "mnist:default,env:MVAE_PAIR_ENC=0" \
  "celeba:lib:base,lib:noxcd,lib:s1only,lib:wgonly,lib:convonly,lib:nobnf,lib:wt64,lib:wt96" \
  "celeba19:lib:base,lib:noxcd,lib:s1only,lib:nobnf" \
  "fashionmnist:lib:base,lib:noxcd,lib:wgonly,lib:wt64,lib:wt96" > $out/ab.txt 2>&1
timeout 60 bash tools/ab_matrix.sh 1 "mnist:default,env:MVAE_PAIR_ENC=0" >> $out/ab.txt 2>&1
TRAFFIC_TABLE=r04_traffic.json timeout 400 bash tools/collect_traffic.sh "convT2d_fwd|4608x128x8x8" "convT2d_fwd|512x128x8x8" "convT2d_wgrad|256x128x4x4" "convT2d_dgrad|512x256x5x5" > $out/traffic.log 2>&1
cp gpurun_out/r04_traffic.json $out/ 2>/dev/null
cat $out/status.txt; cat $out/ab.txt

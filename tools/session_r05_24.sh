#!/bin/bash
# GPU box (MVAE_GIT_HEAD=<head> bash tools/session_r05_24.sh): the PMC traffic table again, complete (the final collection's pass stopped
# at a FashionMNIST key the probe did not know), then the whole GPU suite + smoke on the same head.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s24; rm -rf $out; mkdir -p $out
TRAFFIC_TABLE=r05_traffic.json timeout 420 bash tools/collect_traffic.sh "linear_fwd|M1024 N512 K512" "linear_wgrad_batched|4 layers" "linear_dgrad|M1024 N512 K512" "convT2d_fwd|2048x64x14x14" "convT2d_dgrad|512x256x5x5" "convT2d_dgrad|2048x128x7x7" "convT2d_fwd|4608x128x8x8" "convT2d_wgrad|256x128x4x4" > $out/traffic.log 2> $out/traffic.err
echo "traffic rc=$?" >> $out/status.txt
cp gpurun_out/r05_traffic.json $out/ 2>/dev/null
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$? wall=$(( $(date +%s) - t0 ))s head=$MVAE_GIT_HEAD" >> $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
cat $out/status.txt; tail -4 $out/tests.log; cat $out/traffic.err | tail -5

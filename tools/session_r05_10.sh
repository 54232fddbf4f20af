#!/bin/bash
# round 5, session 10: full GPU suite after the hardware-exp/log Bernoulli term + statistics shift; SQ counters of the CelebA conv launches
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s10; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -12 $out/tests.log >> $out/status.txt
timeout 400 bash tools/collect_sq.sh $out/sq > $out/sq.log 2>&1
cat $out/status.txt; cat $out/sq/sq_counters_pass2.txt | cut -c1-200 | head -50

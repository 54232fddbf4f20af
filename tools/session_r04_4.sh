#!/bin/bash
# GPU box, round 4 session 4: the full GPU suite on the session-3 kernels (after the per-slice BatchNorm fix) + smoke
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s4; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -6 $out/tests.log >> $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
cat $out/status.txt

#!/bin/bash
# GPU box, last seconds of the round: the python-side changes after the final suite (bounded coefficient cache, capture_step state restore)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s27; rm -rf $out; mkdir -p $out
timeout 40 python -m pytest tests/test_capture_step_gpu.py -q -x > $out/capture.log 2>&1; echo "capture rc=$?" > $out/status.txt
timeout 25 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
cat $out/status.txt; tail -3 $out/capture.log

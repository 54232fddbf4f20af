#!/bin/bash
# GPU box, round 4 session 2: the new tests, per-(call, shape) tables of all four steps, one default bench line
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s2; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_celeba19_gpu.py -m gpu -q -k "paired_encoder or loss_bearing or live_oracle" > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -4 $out/tests.log >> $out/status.txt
timeout 500 bash tools/collect_by_shape.sh > $out/by_shape.log 2>&1
cp gpurun_out/by_shape/r04_* $out/ 2>/dev/null
t0=$(date +%s); timeout 500 python bench.py > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s" >> $out/status.txt
cat $out/status.txt; tail -3 $out/bench.err

#!/bin/bash
# GPU box: the step-level subset of collect_all.sh for a change that touches the engine's scheduling but no
# kernel: default bench, per-workload kernel summaries, smoke, and every GPU test but the per-kernel file
# (of which only the optimizer tests run).            -> gpurun_out/final/
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final; rm -rf $out; mkdir -p $out
timeout 100 python bench.py > $out/r02_bench_default.json 2> $out/bench.err; echo "bench rc=$?" > $out/status.txt
bash tools/collect_profiles.sh > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/r02_*_kernel_stats.txt $out/ 2>/dev/null
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
timeout 30 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k adam > $out/tests_adam.log 2>&1; echo "adam tests rc=$?" >> $out/status.txt
tail -1 $out/tests_adam.log >> $out/status.txt
timeout 140 python -m pytest tests -m gpu -q -x --ignore=tests/test_kernels_gpu.py > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/status.txt
tail -1 $out/tests.log >> $out/status.txt
cat $out/status.txt

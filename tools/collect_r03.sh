#!/bin/bash
# GPU box: the round's final measurements -> gpurun_out/final3/  (copy into profiles/ afterwards)
#   full GPU test suite, smoke, the default bench line (all four workloads), the data-parallel launch path at world
#   size 1 for the four workloads (both transports for MNIST), rocprofv3 kernel summaries.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final3; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
grep -E "passed|failed|^FAILED" $out/tests.log >> $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
t0=$(date +%s); timeout 400 python bench.py > $out/r03_bench_default.json 2> $out/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s" >> $out/status.txt
for w in mnist fashionmnist celeba celeba19; do
    timeout 200 python bench.py --workload $w --force-dp --no-extras > $out/dp_$w.json 2> $out/dp_$w.err; echo "dp $w rc=$?" >> $out/status.txt
done
MVAE_COMM=torch timeout 200 python bench.py --workload mnist --force-dp --no-extras > $out/dp_mnist_torch.json 2>/dev/null
bash tools/collect_profiles.sh > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/r03_*_kernel_stats.txt $out/ 2>/dev/null
cat $out/status.txt

#!/bin/bash
# GPU box, round 4 final collection: full GPU suite + smoke, the default bench line (all four workloads), kernel-trace
# summaries + step timelines, per-(call, shape) tables, the data-parallel launch path at world size 1, PMC traffic of the
# dominant launches.  Everything lands under gpurun_out/final4/ (copy into profiles/).
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final4; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -4 $out/tests.log >> $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
# the per-(call, shape) tables first: the bench line reads roofline.rocprof_avg_us from profiles/r04_by_shape.json
timeout 500 bash tools/collect_by_shape.sh > $out/by_shape.log 2>&1
cp gpurun_out/by_shape/r04_* $out/ 2>/dev/null
cp gpurun_out/by_shape/r04_by_shape.json profiles/r04_by_shape.json 2>/dev/null
t0=$(date +%s); timeout 500 python bench.py > $out/r04_bench_default.json 2> $out/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s" >> $out/status.txt
for w in mnist fashionmnist celeba celeba19; do
    timeout 200 python bench.py --workload $w --force-dp --no-extras > $out/dp_$w.json 2> $out/dp_$w.err; echo "dp $w rc=$?" >> $out/status.txt
done
MVAE_COMM=torch timeout 200 python bench.py --workload mnist --force-dp --no-extras > $out/dp_mnist_torch.json 2>/dev/null
timeout 500 bash tools/collect_profiles.sh > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/r04_* $out/ 2>/dev/null
TRAFFIC_TABLE=r04_traffic.json timeout 500 bash tools/collect_traffic.sh "linear_dgrad|M1024 N512 K512" "convT2d_fwd|4608x128x8x8" "convT2d_fwd|512x128x8x8" "convT2d_wgrad|256x128x4x4" "convT2d_dgrad|512x256x5x5" "convT2d_dgrad|2048x128x7x7" > $out/traffic.log 2>&1
cp gpurun_out/r04_traffic.json $out/ 2>/dev/null
cat $out/status.txt

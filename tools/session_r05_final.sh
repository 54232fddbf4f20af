#!/bin/bash
# GPU box, round 5 final collection (run as: MVAE_GIT_HEAD=<head> bash tools/session_r05_final.sh): the per-(call, shape) rocprofv3
# tables + kernel-trace summaries + step timelines of all four workloads, PMC traffic of the dominant launches, the block-count
# quantisation tables, the default bench line (quoting the tables collected HERE, same code), the data-parallel launch path at
# world size 1.  Everything lands under gpurun_out/final5/ (copy into profiles/).
cd "$GRAFT_REPO_ROOT"
export PFX=r05
out=gpurun_out/final5; rm -rf $out; mkdir -p $out
echo "head=$MVAE_GIT_HEAD" > $out/status.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
timeout 600 bash tools/collect_by_shape.sh > $out/by_shape.log 2>&1
cp gpurun_out/by_shape/r05_* $out/ 2>/dev/null
cp gpurun_out/by_shape/r05_by_shape.json profiles/r05_by_shape.json 2>/dev/null
TRAFFIC_TABLE=r05_traffic.json timeout 600 bash tools/collect_traffic.sh "linear_fwd|M1024 N512 K512" "linear_wgrad_batched|4 layers" "linear_dgrad|M1024 N512 K512" "convT2d_fwd|2048x64x14x14" "convT2d_dgrad|512x256x5x5" "convT2d_dgrad|2048x128x7x7" "convT2d_fwd|4608x128x8x8" "convT2d_wgrad|256x128x4x4" > $out/traffic.log 2>&1
cp gpurun_out/r05_traffic.json $out/ 2>/dev/null; cp gpurun_out/r05_traffic.json profiles/r05_traffic.json 2>/dev/null
t0=$(date +%s); timeout 600 python bench.py > $out/r05_bench_default.json 2> $out/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s" >> $out/status.txt
timeout 600 bash tools/collect_profiles.sh > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/r05_* $out/ 2>/dev/null
for w in mnist fashionmnist celeba celeba19; do
    timeout 200 python bench.py --workload $w --force-dp --no-extras > $out/dp_$w.json 2> $out/dp_$w.err; echo "dp $w rc=$?" >> $out/status.txt
    MVAE_GRID_REPORT=1 MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning.so timeout 200 python tools/grid_report.py run $w 2> $out/grid_$w.err > /dev/null
    python tools/grid_report.py table $out/grid_$w.err > $out/grid_$w.txt; rm -f $out/grid_$w.err
done
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning.so timeout 120 python tools/wgrad_probe.py final > $out/wgrad_probe.txt 2>&1
cat $out/status.txt; tail -c 1500 $out/bench.err

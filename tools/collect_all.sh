#!/bin/bash
# GPU box: everything profiles/r02_* is made of, in one call -> gpurun_out/final/
#   tools/collect_all.sh [notests]
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final; rm -rf $out; mkdir -p $out
if [ "$1" != "notests" ]; then
    timeout 900 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
    grep -E "passed|failed" $out/tests.log | tail -2 >> $out/status.txt
    timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/status.txt
fi
timeout 400 python bench.py > $out/r02_bench_default.json 2> $out/bench.err; echo "bench rc=$?" >> $out/status.txt
bash tools/collect_profiles.sh > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/r02_*_kernel_stats.txt $out/ 2>/dev/null
python tools/gemm_bench.py --cases all 2>&1 | grep -v amdgpu.ids > $out/r02_gemm_bench.txt
rm -f gpurun_out/r02_traffic.json
bash tools/collect_traffic.sh "linear_dgrad|M1024 N512 K512" "linear_fwd|M1024 N512 K512" "linear_wgrad|M1024 N512 K512" \
     "convT2d_dgrad|512x256x5x5" "convT2d_fwd|512x128x8x8" "convT2d_wgrad|256x128x4x4" > $out/traffic.log 2>&1
cp gpurun_out/r02_traffic.json $out/ 2>/dev/null
bash tools/collect_sq.sh gpurun_out/sq > $out/sq.log 2>&1
cat gpurun_out/sq/sq_counters_pass1.txt gpurun_out/sq/sq_counters_pass2.txt > $out/r02_celeba_sq_counters.txt 2>/dev/null
tools/bin/mfma_peak > $out/r02_mfma_peak.txt 2>&1
cat $out/status.txt

#!/bin/bash
# GPU box: the acceptance run of the statistics-launch path -- its tests, then an interleaved A/B of
# the CelebA and CelebA-19 steps with and without it, then the kernel summary of the CelebA-19 step with it.
#   gpurun --timeout 420 -- 'bash tools/try_fused_bn_stats.sh'        -> gpurun_out/fused_bn/
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fused_bn; rm -rf $out; mkdir -p $out
MVAE_EXPERIMENTAL=1 timeout 240 python -m pytest tests/test_fused_bn_stats_gpu.py -m gpu -q > $out/tests.log 2>&1
echo "tests rc=$?" > $out/status.txt; tail -3 $out/tests.log >> $out/status.txt
if grep -q "failed\|error" $out/status.txt; then cat $out/status.txt; grep -E "^(FAILED|ERROR|E  )" $out/tests.log | head -40; exit 1; fi
bash tools/ab_env.sh celeba MVAE_FUSED_BN_STATS=1 3 >> $out/status.txt 2>&1
bash tools/ab_env.sh celeba19 MVAE_FUSED_BN_STATS=1 3 >> $out/status.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MVAE_FUSED_BN_STATS=1 rocprofv3 --kernel-trace --stats -d $out/raw -o c19 -- python bench.py --workload celeba19 --no-extras --steps 10 --warmup 5 > $out/c19.log 2>&1
f=$(find $out/raw -name "*.db" | head -1)
[ -n "$f" ] && python tools/rocpd_summary.py "$f" > $out/celeba19_fused_kernel_stats.txt && rm -rf $out/raw
cat $out/status.txt

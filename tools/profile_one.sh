#!/bin/bash
# GPU box: rocprofv3 kernel-trace summary of one bench workload -> gpurun_out/$OUT/<tag>_kernel_stats.txt
# usage: tools/profile_one.sh <workload> <steps> <outdir> <tag> [env assignments...]
set -e
w=$1; steps=$2; out=$3; tag=$4; shift 4
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p $out
env "$@" rocprofv3 --kernel-trace --stats -d $out/raw_$tag -o $tag -- python bench.py --workload $w --no-extras --steps $steps --warmup 5 > $out/$tag.log 2>&1 || true
f=$(find $out/raw_$tag -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" > $out/${tag}_kernel_stats.txt
python tools/rocpd_summary.py "$f" --timeline >> $out/${tag}_kernel_stats.txt
grep metric $out/$tag.log | cut -c1-200 >> $out/${tag}_kernel_stats.txt
rm -rf $out/raw_$tag

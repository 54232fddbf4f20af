#!/bin/bash
# GPU box: rocprofv3 kernel-trace summary of bench.py for "workload steps" pairs (SPECS="celeba 30;mnist 100")
# -> gpurun_out/prof_now/<workload>_stats.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_now; rm -rf $out; mkdir -p $out
IFS=";" read -ra SP <<< "${SPECS:-celeba 30;mnist 100}"
for spec in "${SP[@]}"; do
    set -- $spec; w=$1; steps=$2
    rocprofv3 --kernel-trace --stats -d $out/raw_$w -o $w -- python bench.py --workload $w --no-extras --steps $steps --warmup 5 > $out/$w.log 2>&1
    f=$(find $out/raw_$w -name "*.db" | head -1)
    python tools/rocpd_summary.py "$f" > $out/${w}_stats.txt
    python tools/rocpd_summary.py "$f" --timeline >> $out/${w}_stats.txt
    grep metric $out/$w.log | cut -c1-200 >> $out/${w}_stats.txt
    rm -rf $out/raw_$w
done

#!/bin/bash
# GPU box: rocprofv3 kernel-trace summaries of the bench command for each workload -> gpurun_out/profiles_new/
# (copy the ${PFX}_* files into profiles/)
set -e
PFX=${PFX:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_new
rm -rf $out; mkdir -p $out
only="$*"      # optional: workloads to collect (default all four)
for spec in "mnist 512 100" "celeba 256 30" "celeba19 256 10" "fashionmnist 1024 30"; do
    set -- $spec
    w=$1; b=$2; steps=$3
    if [ -n "$only" ] && ! echo " $only " | grep -q " $w "; then continue; fi
    rocprofv3 --kernel-trace --stats -d $out/raw_$w -o $w -- python bench.py --workload $w --no-extras --steps $steps --warmup 5 > $out/$w.log 2>&1
    f=$(find $out/raw_$w -name "*.db" | head -1)
    python tools/rocpd_summary.py "$f" > $out/${PFX}_${w}_b${b}_kernel_stats.txt
    python tools/rocpd_summary.py "$f" --timeline >> $out/${PFX}_${w}_b${b}_kernel_stats.txt
    grep metric $out/$w.log | cut -c1-200 >> $out/${PFX}_${w}_b${b}_kernel_stats.txt
    python tools/rocpd_summary.py "$f" --step > $out/${PFX}_${w}_step_timeline.txt
    rm -rf $out/raw_$w
done
ls -la $out

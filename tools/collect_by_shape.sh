#!/bin/bash
# GPU box: per-(launcher call, shape) rocprofv3 tables of one step of each workload
#   -> gpurun_out/by_shape/${PFX}_<workload>_by_shape.txt + gpurun_out/by_shape/${PFX}_by_shape.json (copy into profiles/)
PFX=${PFX:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/by_shape; rm -rf $out; mkdir -p $out
for w in "${@:-mnist fashionmnist celeba celeba19}"; do
  for k in $w; do
    rocprofv3 --kernel-trace -d $out/t_$k -o t -- python tools/step_by_shape.py run $k $out/calls_$k.json > $out/$k.run.log 2>&1
    db=$(find $out/t_$k -name "*.db" | head -1)
    python tools/step_by_shape.py collect $out/calls_$k.json "$db" $out/${PFX}_by_shape.json > $out/${PFX}_${k}_by_shape.txt 2> $out/$k.collect.log
    python tools/rocpd_summary.py "$db" > $out/${PFX}_${k}_eager_kernel_stats.txt 2>/dev/null
    rm -rf $out/t_$k
  done
done
ls -la $out

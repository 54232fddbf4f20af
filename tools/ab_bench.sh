#!/bin/bash
# GPU box: ms/step of a workload under two libraries, interleaved: tools/ab_bench.sh <workload> <libB.so> [reps]
w=$1; libb=$2; reps=${3:-3}
for i in $(seq $reps); do
  a=$(python bench.py --workload $w --no-extras 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  b=$(MVAE_HIP_LIB=$libb python bench.py --workload $w --no-extras 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$w  default $a   $(basename $libb) $b"
done

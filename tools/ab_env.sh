#!/bin/bash
# GPU box: ms/step of a workload with and without one environment switch, interleaved:
#   tools/ab_env.sh <workload> <VAR=value> [reps]
w=$1; sw=$2; reps=${3:-3}
ms() { python -c "import sys,json; print(json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])['ms_per_step'])"; }
for i in $(seq $reps); do
    a=$(python bench.py --workload $w --no-extras 2>/dev/null | ms)
    b=$(env "$sw" python bench.py --workload $w --no-extras 2>/dev/null | ms)
    echo "$w  default $a   $sw $b"
done

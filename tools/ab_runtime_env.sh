#!/bin/bash
# GPU box: ms/step of a workload under HIP runtime environment switches, one run each, the default re-measured
# between them:  tools/ab_runtime_env.sh <workload> VAR=value [VAR=value ...]
w=$1; shift
ms() { python -c "import sys,json; l=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; print(json.loads(l[-1])['ms_per_step'] if l else 'failed')"; }
i=0
for sw in "$@"; do
    if [ $((i % 3)) -eq 0 ]; then echo "$w  default  $(timeout 120 python bench.py --workload $w --no-extras 2>/dev/null | ms)"; fi
    echo "$w  $sw  $(timeout 120 env $sw python bench.py --workload $w --no-extras 2>/dev/null | ms)"
    i=$((i + 1))
done

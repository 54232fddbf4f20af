#!/bin/bash
# GPU box: ms/step of workloads under several builds / environment switches, interleaved round by round.
#   tools/ab_matrix.sh <reps> "<workload>:<variant>,<variant>,..." ...
# variant = default | lib:<name> (multimodal-vae-public_amd/libmvae_hip_tuning_<name>.so) | env:VAR=VAL[+VAR=VAL]
# prints one line per (round, workload) with the variants' ms/step side by side.
reps=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
one() {   # workload variant -> ms_per_step
    local w=$1 v=$2 envs=""
    case "$v" in
        default) ;;
        lib:*) envs="MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_${v#lib:}.so" ;;
        env:*) envs=$(echo "${v#env:}" | tr '+' ' ') ;;
    esac
    env $envs timeout 300 python bench.py --workload $w --no-extras 2>/dev/null | python -c "
import sys, json
try:
    print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])
except Exception:
    print('FAIL')"
}
for r in $(seq $reps); do
    for spec in "$@"; do
        w=${spec%%:*}; vs=${spec#*:}
        line="$w"
        IFS=',' read -ra arr <<< "$vs"
        for v in "${arr[@]}"; do line="$line  $v $(one $w $v)"; done
        echo "$line"
    done
done

#!/bin/bash
# HERE (not on the GPU box): copy what `tools/session.sh final <pfx>` left under gpurun_out/final/ into profiles/ -- the tables
# bench.py quotes, the kernel-trace summaries and timelines, the grid report, the world-1 data-parallel lines, the bench line.
#   bash tools/install_final.sh r06 <head>
set -u
cd "$(dirname "$0")/.."
pfx=${1:-r06}; head=${2:-$(git rev-parse --short HEAD)}
src=gpurun_out/final
[ -f $src/status.txt ] || { echo "no $src/status.txt" >&2; exit 1; }
cat $src/status.txt
for f in $src/${pfx}_*; do
    case "$f" in *_eager_kernel_stats.txt) continue ;; esac
    [ -s "$f" ] && cp "$f" profiles/
done
{
    echo "# head $head: blocks per launch against the 256 CUs (tools/grid_report.py, tuning build), all four workloads"
    for w in mnist fashionmnist celeba celeba19; do echo "## $w"; cat $src/grid_$w.txt; done
} > profiles/${pfx}_grid_report.txt
{
    echo "# head $head: bench.py --workload W --force-dp --no-extras at world size 1 (the one-graph mvae_comm transport)"
    for w in mnist fashionmnist celeba celeba19; do
        python - "$src/dp_$w.json" "$w" <<'EOF'
import json, sys
try:
    line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
    d = json.loads(line)
    print('%s %.4f ms/step %.1f images/sec %s' % (sys.argv[2], d['ms_per_step'], d['value'], d['config']['parallelism']))
except Exception as e:
    print('%s: no line (%s)' % (sys.argv[2], e))
EOF
    done
} > profiles/${pfx}_dp_world1.txt
cat profiles/${pfx}_dp_world1.txt
python tools/aggregates.py 2>/dev/null | tail -6

#!/bin/bash
# GPU box: PMC passes for the calls bench.py may name as dominant -> profiles/r03_traffic.json
# usage: tools/collect_traffic.sh "<name>|<key>" ...
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/traffic
rm -rf $out; mkdir -p $out
for spec in "$@"; do
    name="${spec%%|*}"; key="${spec#*|}"
    tag=$(echo "$name $key" | tr ' ' '_')
    rocprofv3 --pmc FETCH_SIZE -d $out/f_$tag -o p -- python tools/traffic_probe.py run "$name" "$key" > $out/$tag.f.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $out/w_$tag -o p -- python tools/traffic_probe.py run "$name" "$key" > $out/$tag.w.log 2>&1
    f=$(find $out/f_$tag -name "*.db" | head -1); w=$(find $out/w_$tag -name "*.db" | head -1)
    python tools/traffic_probe.py collect "$f" "$w" gpurun_out/r03_traffic.json "$name" "$key" > $out/$tag.json
    rm -rf $out/f_$tag $out/w_$tag
done
cat gpurun_out/r03_traffic.json

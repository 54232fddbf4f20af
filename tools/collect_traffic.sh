#!/bin/bash
# GPU box: PMC passes for the calls bench.py may name as dominant -> gpurun_out/<table>.json (copy into profiles/)
# usage: TRAFFIC_TABLE=r04_traffic.json tools/collect_traffic.sh "<name>|<key>" ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
table=${TRAFFIC_TABLE:-r06_traffic.json}
out=gpurun_out/traffic
rm -rf $out; mkdir -p $out
for spec in "$@"; do
    name="${spec%%|*}"; key="${spec#*|}"
    tag=$(echo "$name $key" | tr ' ' '_')
    rocprofv3 --pmc FETCH_SIZE -d $out/f_$tag -o p -- python tools/traffic_probe.py run "$name" "$key" > $out/$tag.f.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $out/w_$tag -o p -- python tools/traffic_probe.py run "$name" "$key" > $out/$tag.w.log 2>&1
    f=$(find $out/f_$tag -name "*.db" | head -1); w=$(find $out/w_$tag -name "*.db" | head -1)
    # a spec the probe does not know (or a failed pass) costs that entry, not the table
    python tools/traffic_probe.py collect "$f" "$w" gpurun_out/$table "$name" "$key" > $out/$tag.json || echo "SKIPPED $spec" >&2
    rm -rf $out/f_$tag $out/w_$tag
done
cat gpurun_out/$table

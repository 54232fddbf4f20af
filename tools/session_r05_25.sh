#!/bin/bash
# GPU box, the round's last minutes: (1) L2 hit rate and fabric read requests (all / 32-B) of FashionMNIST's convT forward 7x7 -> 14x14,
# of its 8x8 -> 16x16 sibling in CelebA-19 and of CelebA's 16x16 -> 32x32 pair kernel -- is the 3.3x traffic of the first an L2 that
# does not hold the input between the four parity classes?  (2) one driver-style bench run on the final head.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s25; rm -rf $out; mkdir -p $out
for spec in "convT2d_fwd|2048x64x14x14" "convT2d_fwd|4608x64x16x16" "convT2d_fwd|512x32x32x32"; do
    name="${spec%%|*}"; key="${spec#*|}"
    for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
        rm -rf $out/raw
        timeout 100 rocprofv3 --pmc $pass -d $out/raw -o p -- python tools/traffic_probe.py run "$name" "$key" > $out/run.log 2>&1
        f=$(find $out/raw -name "*.db" | head -1)
        echo "## $name $key   [$pass]" >> $out/l2.txt
        [ -n "$f" ] && python tools/traffic_probe.py counters "$f" >> $out/l2.txt 2>&1 || tail -3 $out/run.log >> $out/l2.txt
    done
done
rm -rf $out/raw
t0=$(date +%s); timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_style.json 2> $out/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s head=$MVAE_GIT_HEAD" >> $out/status.txt
cat $out/l2.txt; cat $out/status.txt

#!/bin/bash
# GPU box, round 4 session 23: texture-addresser / cache counters of the CelebA conv launches (tools/pmc_probe.py run): are the
# gather-fed igemm forms addresser-bound the way the small-channel kernel was?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s23; rm -rf $out; mkdir -p $out
pass() {  # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $out/raw_$n -o p -- python tools/pmc_probe.py run > $out/run_$n.log 2>&1
  f=$(find $out/raw_$n -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_probe.py show "$f" > $out/sq_$n.txt
  rm -rf $out/raw_$n
}
pass a GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_MFMA TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass b GRBM_GUI_ACTIVE TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_BUSY_avr SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
for n in a b; do echo "== $n"; cut -c1-75,78-400 $out/sq_$n.txt 2>/dev/null | awk 'NR==1 || NR%3==0' | head -16; tail -2 $out/run_$n.log | cut -c1-160; done

#!/bin/bash
# GPU box, round 4 session 20: SQ / cache counters of the <= 4-channel conv launches (what a 67-us launch whose ALU, texture path
# and memory work are each ~20 us is waiting for)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s20; rm -rf $out; mkdir -p $out
pass() {  # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $out/raw_$n -o p -- python tools/pmc_probe.py run-small > $out/run_$n.log 2>&1
  f=$(find $out/raw_$n -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_probe.py show "$f" > $out/sq_$n.txt
  rm -rf $out/raw_$n
}
pass a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM
pass c TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
ls -la $out; for n in a b c; do echo "== $n"; cut -c1-60,78-400 $out/sq_$n.txt 2>/dev/null | awk 'NR==1 || NR%3==0' | head -20; tail -3 $out/run_$n.log | cut -c1-200; done

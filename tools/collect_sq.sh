#!/bin/bash
# GPU box: SQ counters of the conv launches of the CelebA B=256 step (tools/pmc_probe.py run) -> $1
set -e
out=${1:-gpurun_out/sq}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/raw1 -o p -- python tools/pmc_probe.py run > $out/run1.log 2>&1
f=$(find $out/raw1 -name "*.db" | head -1)
python tools/pmc_probe.py show "$f" > $out/sq_counters_pass1.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES -d $out/raw2 -o p -- python tools/pmc_probe.py run > $out/run2.log 2>&1 || true
f=$(find $out/raw2 -name "*.db" | head -1)
[ -n "$f" ] && python tools/pmc_probe.py show "$f" > $out/sq_counters_pass2.txt
rm -rf $out/raw1 $out/raw2

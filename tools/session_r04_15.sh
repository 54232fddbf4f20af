#!/bin/bash
# GPU box, round 4 session 15: instruction counts (SQ counters) of the CelebA conv launches with and without their epilogue
# stores (MVAE_KO_EPI build) -- what the 40 us of the 32-row kernels' stores are made of
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s15; rm -rf $out; mkdir -p $out
for v in default koepi; do
  lib=""; [ $v = koepi ] && lib="$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_koepi.so"
  MVAE_HIP_LIB=$lib rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES -d $out/raw_$v -o p -- python tools/pmc_probe.py run > $out/run_$v.log 2>&1
  f=$(find $out/raw_$v -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_probe.py show "$f" > $out/sq_$v.txt
  MVAE_HIP_LIB=$lib rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD -d $out/raw2_$v -o p -- python tools/pmc_probe.py run > $out/run2_$v.log 2>&1
  f=$(find $out/raw2_$v -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_probe.py show "$f" > $out/sq2_$v.txt
  rm -rf $out/raw_$v $out/raw2_$v
done
ls -la $out; grep -E "EpNCHWPa" $out/sq_default.txt | head -3 | cut -c1-220; grep -E "EpNCHWPa" $out/sq_koepi.txt | head -3 | cut -c1-220

#!/bin/bash
# GPU box, round 4 session 12: where the time of the conv launches goes -- tools/gemm_bench.py on the CelebA layer shapes
# under knock-out builds: koepi = epilogues store nothing, ko4 = main loop without loads / LDS / barriers (MFMAs, set-up and
# epilogue only), ko4epi = both (MFMAs + set-up only).  Results of the knock-out builds are wrong by construction.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s16; rm -rf $out; mkdir -p $out
for v in base koepi ko4 ko4epi; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/conv_$v.txt 2>&1
  echo "$v rc=$?" >> $out/status.txt
done
cat $out/status.txt
python - <<'P'
import re
vs=['base','koepi','ko4','ko4epi']; t={}
for v in vs:
    for line in open('gpurun_out/s16/conv_%s.txt'%v):
        m=re.match(r'(.{34}) +([\d.]+) \| +([\d.nan]+) +([\d.]+) us', line)
        if m: t.setdefault(m.group(1).strip(),{})[v]=(float(m.group(2)), float(m.group(4)))
print('%-34s %7s %8s | %8s %8s %8s %8s   (us per launch, hot re-issue)' % ('op','GFLOP','mfma us','base','koepi','ko4','ko4epi'))
for k,d in t.items():
    if len(d)==4: print('%-34s %7.2f %8.1f | %8.1f %8.1f %8.1f %8.1f' % (k, d['base'][0], d['base'][0]/157.3*1e3/1e3*1e0 if False else d['base'][0]/157.3*1e3, d['base'][1], d['koepi'][1], d['ko4'][1], d['ko4epi'][1]))
P
# and the instruction counts of the same builds (the first koepi build had deleted its MFMAs: SQ_INSTS_MFMA must be equal here)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in base koepi; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES -d $out/raw_$v -o p -- python tools/pmc_probe.py run > $out/run_$v.log 2>&1
  f=$(find $out/raw_$v -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_probe.py show "$f" > $out/sq_$v.txt
  rm -rf $out/raw_$v
done
grep -E "EpNCHWPa" $out/sq_base.txt | head -2 | cut -c1-220; grep -E "EpNCHWPa" $out/sq_koepi.txt | head -2 | cut -c1-220

#!/bin/bash
# GPU box, round 4 session 12: where the time of the conv launches goes -- tools/gemm_bench.py on the CelebA layer shapes
# under knock-out builds: koepi = epilogues store nothing, ko4 = main loop without loads / LDS / barriers (MFMAs, set-up and
# epilogue only), ko4epi = both (MFMAs + set-up only).  Results of the knock-out builds are wrong by construction.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s12; rm -rf $out; mkdir -p $out
for v in base koepi ko4 ko4epi; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/conv_$v.txt 2>&1
  echo "$v rc=$?" >> $out/status.txt
done
cat $out/status.txt
python - <<'P'
import re
vs=['base','koepi','ko4','ko4epi']; t={}
for v in vs:
    for line in open('gpurun_out/s12/conv_%s.txt'%v):
        m=re.match(r'(.{34}) +([\d.]+) \| +([\d.nan]+) +([\d.]+) us', line)
        if m: t.setdefault(m.group(1).strip(),{})[v]=(float(m.group(2)), float(m.group(4)))
print('%-34s %7s %8s | %8s %8s %8s %8s   (us per launch, hot re-issue)' % ('op','GFLOP','mfma us','base','koepi','ko4','ko4epi'))
for k,d in t.items():
    if len(d)==4: print('%-34s %7.2f %8.1f | %8.1f %8.1f %8.1f %8.1f' % (k, d['base'][0], d['base'][0]/157.3*1e3/1e3*1e0 if False else d['base'][0]/157.3*1e3, d['base'][1], d['koepi'][1], d['ko4'][1], d['ko4epi'][1]))
P

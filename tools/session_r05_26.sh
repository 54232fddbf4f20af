#!/bin/bash
# GPU box, the round's last session: two remedies for the class re-reads / shared output lines of the transposed-conv forward family
# (profiles/r05_convT_l2_counters.txt), as tuning variants built from tools/patches/convT_class_experiments.patch:
#   nt   -DMVAE_EP_NT=2       non-temporal buffer stores in the NCHW epilogues
#   pn   -DMVAE_PAIR_NEIGH=1  the two class pairs of a j tile on neighbouring blocks of ONE XCD (2 items per block) instead of 4 items of one block
#   pnnt both
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s26; rm -rf $out; mkdir -p $out
L=$PWD/multimodal-vae-public_amd
for v in base nt pn pnnt; do
    MVAE_HIP_LIB=$L/libmvae_hip_tuning_$v.so timeout 40 python tools/convT_class_probe.py $v >> $out/probe.txt 2>> $out/probe.err
done
V="lib:base,lib:nt,lib:pn,lib:pnnt"
timeout 110 bash tools/ab_matrix.sh 2 "fashionmnist:$V" "celeba19:$V" "celeba:$V" > $out/ab.txt 2>&1
cat $out/probe.txt $out/ab.txt; tail -3 $out/probe.err

#!/bin/bash
# round 5, session 14: the 64x64-wave-tile batch kernel at 121 registers (one chunk per register set): does it share a CU with the chain kernel?
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s14; rm -rf $out; mkdir -p $out
timeout 600 bash tools/ab_matrix.sh 4 "mnist:lib:base,lib:pd1" > $out/ab_mnist.txt 2>&1
for v in base pd1; do MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 120 python tools/wgrad_probe.py $v >> $out/probe.txt 2>> $out/probe.err; done
cat $out/ab_mnist.txt $out/probe.txt

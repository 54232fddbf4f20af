#!/bin/bash
# GPU box: the conv knock-out table on the round-5 kernels, with the knock-outs ALSO applied to the multi-item loops
# (round 4's MVAE_KO guards sat only in the single-item loops: its "MFMA only" rows for the K <= 256 forms still loaded, staged and
# synchronised).  Builds: tools/build_variants.sh ko1 -DMVAE_KO=1 ... ko4epi "-DMVAE_KO=4 -DMVAE_KO_EPI=1" koepi -DMVAE_KO_EPI=1.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s23; rm -rf $out; mkdir -p $out
L=$PWD/multimodal-vae-public_amd
for v in base ko1 ko3 ko4 ko4epi koepi; do
    lib=$L/libmvae_hip_tuning_$v.so; [ $v = base ] && lib=$L/libmvae_hip_tuning.so
    MVAE_HIP_LIB=$lib timeout 120 python tools/gemm_bench.py --cases conv --auto-only > $out/gemm_$v.txt 2> $out/gemm_$v.err
    echo "$v rc=$?" >> $out/status.txt
done
cat $out/status.txt

#!/bin/bash
# round 5, session 11: MNIST -- PHASED_PRELOAD 1 / 2 settled (x3 interleaved), the step's scheduling switches re-checked on the new batch kernel
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s11; rm -rf $out; mkdir -p $out
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_pl2.so timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x -k "linear or mnist" > $out/tests_pl2.log 2>&1; echo "pl2 tests rc=$?" > $out/status.txt
tail -3 $out/tests_pl2.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 "mnist:lib:base,lib:pl1,lib:pl2,env:MVAE_WGRAD_SIDE=0,env:MVAE_PAIR_ENC=1,env:MVAE_MAIN_FIRST_DEC=1" > $out/ab_mnist.txt 2>&1
cat $out/status.txt $out/ab_mnist.txt

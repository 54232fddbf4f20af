#!/bin/bash
# round 5, session 3: what bounds the batched Linear weight gradient -- the launch alone (hot / cold) under the round-4 kernel, the
# v2 wave-tile shapes and the knock-outs; chain-kernel wave priority in the MNIST step
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s3; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x -k "wgrad or update_their" > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -3 $out/tests.log >> $out/status.txt
for v in base old s21 s11 komfma koload; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 120 python tools/wgrad_probe.py $v >> $out/probe.txt 2>> $out/probe.err
done
timeout 900 bash tools/ab_matrix.sh 3 "mnist:lib:base,lib:old,lib:prio,lib:prio_old" > $out/ab_mnist.txt 2>&1
cat $out/status.txt $out/probe.txt $out/ab_mnist.txt

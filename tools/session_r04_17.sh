#!/bin/bash
# GPU box, round 4 session 17: Linear weight-gradient batches that apply Adam to their own outputs (MVAE_FUSE_ADAM):
# kernel + engine tests, step A/B on MNIST (default on) and forced on the conv models
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s17; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "wgrad_batched or adam" > $out/t_kern.log 2>&1; echo "kernel tests rc=$?" > $out/status.txt
tail -3 $out/t_kern.log >> $out/status.txt
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_replay_parity_gpu.py -m gpu -q -x > $out/t_eng.log 2>&1; echo "engine tests rc=$?" >> $out/status.txt
tail -3 $out/t_eng.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 \
  "mnist:default,env:MVAE_FUSE_ADAM=0" > $out/ab.txt 2>&1
timeout 600 bash tools/ab_matrix.sh 2 \
  "fashionmnist:default,env:MVAE_FUSE_ADAM=1" \
  "celeba:default,env:MVAE_FUSE_ADAM=1" >> $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

#!/bin/bash
# GPU box, round 4 session 18: fused-Adam weight-gradient batches with the parameter / moments prefetched ahead of the reduction,
# the step-counter launch on the label DECODER's branch: tests, then MNIST A/B over both switches (and the conv models)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s18; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "wgrad_batched or adam" > $out/t_kern.log 2>&1; echo "kernel tests rc=$?" > $out/status.txt
tail -3 $out/t_kern.log >> $out/status.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "mnist or update or early or trajectory" > $out/t_eng.log 2>&1; echo "engine tests rc=$?" >> $out/status.txt
tail -3 $out/t_eng.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 \
  "mnist:default,env:MVAE_FUSE_ADAM=0,env:MVAE_COUNTER_RIDES=encoder,env:MVAE_FUSE_ADAM=0+MVAE_COUNTER_RIDES=encoder" > $out/ab.txt 2>&1
timeout 600 bash tools/ab_matrix.sh 2 \
  "fashionmnist:default,env:MVAE_COUNTER_RIDES=encoder" \
  "celeba:default,env:MVAE_COUNTER_RIDES=encoder" >> $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

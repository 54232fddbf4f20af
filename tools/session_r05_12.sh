#!/bin/bash
# round 5, session 12: PHASED_PRELOAD=2 as the default (all four workloads against =0), decoders' Adam early (MVAE_SPLIT_ADAM=1)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s12; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -q -x -k "updated_early or linear or mnist or trajectory" > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -3 $out/tests.log >> $out/status.txt
timeout 600 bash tools/ab_matrix.sh 3 "mnist:default,env:MVAE_SPLIT_ADAM=1,lib:base,lib:pl0" > $out/ab_mnist.txt 2>&1
timeout 900 bash tools/ab_matrix.sh 2 "celeba:default,env:MVAE_SPLIT_ADAM=1,lib:base,lib:pl0" "fashionmnist:default,env:MVAE_SPLIT_ADAM=1,lib:base,lib:pl0" "celeba19:lib:base,lib:pl0" > $out/ab_conv.txt 2>&1
cat $out/status.txt $out/ab_mnist.txt $out/ab_conv.txt

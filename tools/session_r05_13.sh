#!/bin/bash
# round 5, session 13: four k-tiles in flight in the k-grouped layouts (MVAE_PHASED_DEPTH=4) -- parity, step A/B x3, MNIST per-(call, shape) table
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s13; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_celeba19_gpu.py -q -x -k "linear or mnist or grouped or attr or celeba19" > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -3 $out/tests.log >> $out/status.txt
timeout 600 bash tools/ab_matrix.sh 3 "mnist:lib:base,lib:d2" > $out/ab_mnist.txt 2>&1
timeout 600 bash tools/ab_matrix.sh 2 "celeba:lib:base,lib:d2" "fashionmnist:lib:base,lib:d2" "celeba19:lib:base,lib:d2" > $out/ab_conv.txt 2>&1
PFX=r05 timeout 300 bash tools/collect_by_shape.sh mnist > $out/by_shape.log 2>&1
cp gpurun_out/by_shape/r05_mnist_by_shape.txt $out/ 2>/dev/null
cat $out/status.txt $out/ab_mnist.txt $out/ab_conv.txt; head -12 $out/r05_mnist_by_shape.txt | cut -c1-120

#!/bin/bash
# round 5, session 2: the v2 batched Linear weight-gradient kernel -- parity, step A/B against the round-4 kernel and the
# forced wave-tile shapes, per-(call, shape) table + step timeline of MNIST, fabric traffic of the launch
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s2; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x -k "wgrad or linear or mnist or fashion" > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -3 $out/tests.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 "mnist:lib:base,lib:old,lib:s21,lib:s11" > $out/ab_mnist.txt 2>&1
timeout 600 bash tools/ab_matrix.sh 2 "celeba:lib:base,lib:old" "fashionmnist:lib:base,lib:old" "celeba19:lib:base,lib:old" > $out/ab_conv.txt 2>&1
PFX=r05 timeout 300 bash tools/collect_by_shape.sh mnist > $out/by_shape.log 2>&1
cp gpurun_out/by_shape/r05_mnist_by_shape.txt $out/ 2>/dev/null
PFX=r05 timeout 300 bash tools/collect_profiles.sh mnist > $out/collect_profiles.log 2>&1
cp gpurun_out/profiles_new/r05_mnist* $out/ 2>/dev/null
TRAFFIC_TABLE=r05_traffic.json timeout 300 bash tools/collect_traffic.sh "linear_wgrad_batched|4 layers" > $out/traffic.log 2>&1
cp gpurun_out/r05_traffic.json $out/ 2>/dev/null
cat $out/status.txt $out/ab_mnist.txt $out/ab_conv.txt

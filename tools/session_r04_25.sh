#!/bin/bash
# GPU box, round 4 session 25: epilogue operands fetched in batches -- the igemm tile epilogue of the Linear launches (eight
# outputs at a time) and conv_small_fwd_kernel's Swish' path (eight channels at a time, one batch ahead) -- against one by one
# under block-uniform branches (nobatch): parity tests, step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s25; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "linear or lin or conv" > $out/t_kern.log 2>&1; echo "kernel tests rc=$?" > $out/status.txt
tail -2 $out/t_kern.log >> $out/status.txt
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "goldens" > $out/t_eng.log 2>&1; echo "engine tests rc=$?" >> $out/status.txt
tail -2 $out/t_eng.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 2 "fashionmnist:lib:nobatch,lib:batch" "celeba:lib:nobatch,lib:batch" "mnist:lib:nobatch,lib:batch" > $out/ab.txt 2>&1
timeout 200 bash tools/ab_matrix.sh 1 "celeba19:lib:nobatch,lib:batch" >> $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

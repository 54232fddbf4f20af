#!/bin/bash
# GPU box, round 4 session 11: shifts instead of run-time divisions in the conv tile set-up / epilogue of power-of-two layers
# (default) against divisions (nofd)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s11; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_conv.log 2>&1; echo "conv tests rc=$?" > $out/status.txt
tail -2 $out/t_conv.log >> $out/status.txt
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_celeba19_gpu.py -m gpu -q -k "golden or baseline" > $out/t_engine.log 2>&1; echo "engine tests rc=$?" >> $out/status.txt
tail -2 $out/t_engine.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 3 \
  "celeba19:lib:base,lib:nofd" \
  "celeba:lib:base,lib:nofd" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

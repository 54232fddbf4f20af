#!/bin/bash
# GPU box, round 4 session 3: statistics-only conv + merge, PoE chunks, per-slice BatchNorm, 32-row decode hoist
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s3; rm -rf $out; mkdir -p $out
timeout 500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "poe or stats or batchnorm or conv_transpose or product" > $out/t_kernels.log 2>&1; echo "kernels rc=$?" > $out/status.txt
tail -3 $out/t_kernels.log >> $out/status.txt
timeout 700 python -m pytest tests/test_celeba19_gpu.py tests/test_replay_parity_gpu.py tests/test_engine_gpu.py -m gpu -q -k "celeba" > $out/t_engine.log 2>&1; echo "engine rc=$?" >> $out/status.txt
tail -3 $out/t_engine.log >> $out/status.txt
timeout 900 bash tools/ab_matrix.sh 2 \
  "celeba19:default,env:MVAE_STATS_CONV=0,lib:base,lib:nochunk,lib:nohoist,lib:noslice" \
  "celeba:default,env:MVAE_STATS_CONV=0,lib:base,lib:nohoist,lib:noslice" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

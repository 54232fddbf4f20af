#!/bin/bash
# GPU box, round 4 session 8: the rank supervisor end to end on a GPU (world size 1 under the real launcher), SQ counters
# of the CelebA conv launches on the final kernels
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s8; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_comm_gpu.py -m gpu -q > $out/t_comm.log 2>&1; echo "comm tests rc=$?" > $out/status.txt
tail -3 $out/t_comm.log >> $out/status.txt
timeout 400 bash tools/collect_sq.sh $out/sq > $out/sq.log 2>&1; echo "sq rc=$?" >> $out/status.txt
cat $out/status.txt

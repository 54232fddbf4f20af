#!/bin/bash
# round 5, session 1: the full GPU suite on the tree with the world-2 communicator test, the zero-gradient bound and celeba19_b8
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s1; rm -rf $out; mkdir -p $out
timeout 420 python -m pytest tests/test_comm_world2_gpu.py -q -x -rA > $out/world2.log 2>&1; echo "world2 rc=$?" > $out/status.txt
tail -15 $out/world2.log >> $out/status.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 --deselect tests/test_comm_world2_gpu.py > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/status.txt
tail -8 $out/tests.log >> $out/status.txt
cat $out/status.txt

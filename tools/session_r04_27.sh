#!/bin/bash
# GPU box, round 4 session 27: the loss-folding epilogues (Linear + Bernoulli / categorical term) with their operands fetched
# ahead of the main loop (rr) against inside the epilogue (norr): parity tests, MNIST step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s27; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "bce or ce or fold or loss" > $out/t_kern.log 2>&1; echo "kernel tests rc=$?" > $out/status.txt
tail -2 $out/t_kern.log >> $out/status.txt
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "mnist and (oracle or golden)" > $out/t_eng.log 2>&1; echo "engine tests rc=$?" >> $out/status.txt
tail -2 $out/t_eng.log >> $out/status.txt
timeout 600 bash tools/ab_matrix.sh 3 "mnist:lib:norr,lib:rr" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt

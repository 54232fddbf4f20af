#!/bin/bash
# GPU box, round 4 session 24: the small layouts' epilogue operands (bias / producer pre-activation / mask) fetched ahead of
# the main loop (pre) against inside the epilogue under block-uniform branches (nopre): Linear parity tests, hot re-issue of
# the MNIST GEMM shapes, step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s24; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "linear or lin" > $out/t_lin.log 2>&1; echo "linear tests rc=$?" > $out/status.txt
tail -2 $out/t_lin.log >> $out/status.txt
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "mnist and (oracle or golden)" > $out/t_eng.log 2>&1; echo "engine tests rc=$?" >> $out/status.txt
tail -2 $out/t_eng.log >> $out/status.txt
for v in nopre pre; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 200 python tools/gemm_bench.py --cases lin --auto-only > $out/lin_$v.txt 2>&1
done
timeout 600 bash tools/ab_matrix.sh 3 "mnist:lib:nopre,lib:pre" > $out/ab.txt 2>&1
timeout 400 bash tools/ab_matrix.sh 1 "fashionmnist:lib:nopre,lib:pre" "celeba:lib:nopre,lib:pre" "celeba19:lib:nopre,lib:pre" >> $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt
paste <(cut -c1-100 $out/lin_nopre.txt) <(cut -c46-100 $out/lin_pre.txt) | head -40

#!/bin/bash
# round 5, session 6: NCHW tile epilogues through buffer stores -- parity, per-launch times (gemm_bench, hot), step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s6; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_celeba19_gpu.py tests/test_capture_step_gpu.py -q -x > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -5 $out/tests.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_base.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/gemm_new.txt 2>&1
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_epold.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/gemm_old.txt 2>&1
timeout 900 bash tools/ab_matrix.sh 2 "celeba:lib:base,lib:epold" "fashionmnist:lib:base,lib:epold" "celeba19:lib:base,lib:epold" > $out/ab_conv.txt 2>&1
cat $out/status.txt; paste $out/gemm_new.txt $out/gemm_old.txt | cut -c1-220; cat $out/ab_conv.txt

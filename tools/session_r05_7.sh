#!/bin/bash
# round 5, session 7: gather loaders with one offset per tap (conv kernels at 4 waves per SIMD) -- parity, per-launch times over the
# tilings, step A/B; the captured-vs-eager CelebA difference
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s7; rm -rf $out; mkdir -p $out
timeout 120 python tools/capture_debug.py celeba 6 > $out/capture_debug.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_celeba19_gpu.py -q -x > $out/tests.log 2>&1; echo "tests rc=$?" > $out/status.txt
tail -5 $out/tests.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning.so timeout 300 python tools/gemm_bench.py --cases conv > $out/gemm_new.txt 2>&1
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_base.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/gemm_base.txt 2>&1
cp multimodal-vae-public_amd/libmvae_hip_tuning.so multimodal-vae-public_amd/libmvae_hip_tuning_new.so
timeout 900 bash tools/ab_matrix.sh 2 "celeba:lib:new,lib:base" "fashionmnist:lib:new,lib:base" "celeba19:lib:new,lib:base" > $out/ab_conv.txt 2>&1
cat $out/status.txt $out/capture_debug.txt; cat $out/gemm_new.txt; tail -3 $out/gemm_base.txt; cat $out/ab_conv.txt

#!/bin/bash
# GPU box, round 4 session 13: NCHW epilogue stores issued through inline asm (outside the compiler's vmcnt bookkeeping;
# default) against plain stores (noasm): conv + engine parity tests, hot re-issue of the CelebA conv launches, step A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s13; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv" > $out/t_conv.log 2>&1; echo "conv tests rc=$?" > $out/status.txt
tail -2 $out/t_conv.log >> $out/status.txt
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_celeba19_gpu.py tests/test_replay_parity_gpu.py -m gpu -q > $out/t_engine.log 2>&1; echo "engine+replay tests rc=$?" >> $out/status.txt
tail -2 $out/t_engine.log >> $out/status.txt
for v in base noasm; do
  MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_$v.so timeout 300 python tools/gemm_bench.py --cases conv --auto-only > $out/conv_$v.txt 2>&1
done
timeout 900 bash tools/ab_matrix.sh 3 \
  "celeba19:lib:base,lib:noasm" \
  "celeba:lib:base,lib:noasm" \
  "fashionmnist:lib:base,lib:noasm" > $out/ab.txt 2>&1
cat $out/status.txt; cat $out/ab.txt
paste <(cut -c1-34,46-100 $out/conv_base.txt) <(cut -c46-100 $out/conv_noasm.txt) | head -30

#!/bin/bash
# round 5, session 5: capture_step tests again; block-count quantisation of every GEMM-shaped launch (celeba, fashionmnist, celeba19)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s5; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_capture_step_gpu.py -q > $out/capture.log 2>&1; echo "capture rc=$?" > $out/status.txt
tail -12 $out/capture.log >> $out/status.txt
for w in celeba fashionmnist celeba19 mnist; do
  MVAE_GRID_REPORT=1 MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning.so timeout 200 python tools/grid_report.py run $w 2> $out/grid_$w.err > /dev/null
  python tools/grid_report.py table $out/grid_$w.err > $out/grid_$w.txt
done
cat $out/status.txt; head -40 $out/grid_celeba.txt

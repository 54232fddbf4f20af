#!/bin/bash
# round 5, session 4: capture_step + module-surface parity at BASELINE batches; wave-tile cost model, batch priority / 16 waves in the step
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/s4; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_capture_step_gpu.py -q -x > $out/capture.log 2>&1; echo "capture rc=$?" > $out/status.txt
tail -25 $out/capture.log >> $out/status.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -k "module_surface or update_their" > $out/surface.log 2>&1; echo "surface rc=$?" >> $out/status.txt
tail -6 $out/surface.log >> $out/status.txt
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_base.so timeout 120 python tools/wgrad_probe.py base >> $out/probe.txt 2>> $out/probe.err
MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning_kw16.so timeout 120 python tools/wgrad_probe.py kw16 >> $out/probe.txt 2>> $out/probe.err
timeout 900 bash tools/ab_matrix.sh 3 "mnist:lib:base,lib:old,lib:bprio,lib:kw16,lib:bpkw16" > $out/ab_mnist.txt 2>&1
timeout 300 python - > $out/module_surface.txt 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
import bench
for kind, batch in (('mnist', 512), ('celeba', 256)):
    print(kind, json.dumps(bench.module_surface(kind, batch, torch.device('cuda', 0))))
PY
cat $out/status.txt $out/probe.txt $out/ab_mnist.txt $out/module_surface.txt

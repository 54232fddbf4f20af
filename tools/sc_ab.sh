cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv" 2>&1 | tail -4
cd tools
for v in _sc2off "" _sc2s3; do
  echo "== variant ${v:-base}"
  MVAE_HIP_LIB=$GRAFT_REPO_ROOT/multimodal-vae-public_amd/libmvae_hip_tuning$v.so python small_conv_probe.py 2>/dev/null | grep wgrad
done

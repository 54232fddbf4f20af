"""GPU: the input pipeline kernels through the C ABI, byte-exact against the Pillow goldens and the
oracle (integer work: equality, no tolerance)."""
import os

import numpy as np
import pytest
import torch

import mvae_amd  # noqa: F401
from mvae_amd import preprocess as PP
from oracle import preprocess as OP

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess.npz')


def test_resize_center_crop_matches_pillow_goldens():
    fx = np.load(GOLD)
    pipe = PP.ResizeCenterCropToTensor(64)
    n = len([k for k in fx.files if k.startswith('image_')])
    for k in range(n):
        img = fx['image_%d' % k]
        out = pipe(torch.from_numpy(img[None]).to(DEV)).cpu().numpy()[0]
        assert np.array_equal(out, fx['final_%d' % k]), 'case %d (%s)' % (k, img.shape)


def test_celeba_batch_matches_oracle():
    """A batch of CelebA-sized images: every image of the launch, every pixel."""
    rng = np.random.RandomState(11)
    batch = rng.randint(0, 256, (9, 218, 178, 3)).astype(np.uint8)
    batch[3] = 255; batch[4] = 0                                   # saturated images stay saturated
    out = PP.ResizeCenterCropToTensor(64)(torch.from_numpy(batch).to(DEV)).cpu().numpy()
    assert out.shape == (9, 3, 64, 64) and out.dtype == np.float32
    for b in range(9):
        assert np.array_equal(out[b], OP.resize_center_crop_to_tensor(batch[b])), 'image %d' % b
    assert (out[3] == 1.0).all() and (out[4] == 0.0).all()


def test_to_tensor():
    rng = np.random.RandomState(2)
    u8 = rng.randint(0, 256, (5, 28, 28)).astype(np.uint8)
    out = PP.to_tensor(torch.from_numpy(u8).to(DEV))
    assert out.shape == (5, 1, 28, 28)
    assert np.array_equal(out.cpu().numpy()[:, 0], u8.astype(np.float32) / np.float32(255.0))
    allv = torch.arange(256, dtype=torch.uint8, device=DEV).reshape(1, 16, 16)
    assert np.array_equal(PP.to_tensor(allv).cpu().numpy().ravel(), np.arange(256, dtype=np.float32) / np.float32(255.0))


def test_rejects_bad_input():
    pipe = PP.ResizeCenterCropToTensor(64)
    with pytest.raises(TypeError):
        pipe(torch.zeros(2, 218, 178, 3, device=DEV))               # float, not uint8
    with pytest.raises(RuntimeError):
        pipe(torch.zeros(2, 218, 178, 3, dtype=torch.uint8))        # not on the GPU

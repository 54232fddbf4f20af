"""CPU: the input-pipeline oracle (numpy restatement of Pillow's 8-bit bilinear resample + torchvision's
size / crop rules, celeba/train.py:146-148) against the committed Pillow outputs, against Pillow live
where it is installed, and the library's HOST coefficient builder against the oracle's."""
import os

import numpy as np
import pytest

import mvae_amd  # noqa: F401
from mvae_amd import preprocess as PP
from oracle import preprocess as OP

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess.npz')


def cases():
    fx = np.load(GOLD)
    n = len([k for k in fx.files if k.startswith('image_')])
    return [(fx['image_%d' % k], fx['resized_%d' % k], fx['final_%d' % k]) for k in range(n)]


def test_oracle_matches_pillow_goldens():
    for img, resized, final in cases():
        h, w, _ = img.shape
        nh, nw = OP.resized_size(h, w, 64)
        assert resized.shape == (nh, nw, 3)
        assert np.array_equal(OP.resize_bilinear_u8(img, nw, nh), resized), 'resize %dx%d' % (h, w)
        got = OP.resize_center_crop_to_tensor(img)
        assert got.dtype == np.float32 and np.array_equal(got, final), 'pipeline %dx%d' % (h, w)


def test_celeba_geometry():
    """218 x 178 aligned-and-cropped CelebA: Resize(64) -> 78 x 64, CenterCrop(64) starts at row 7."""
    assert OP.resized_size(218, 178, 64) == (78, 64)
    assert OP.center_crop_origin(78, 64, 64) == (7, 0)
    assert OP.resized_size(100, 160, 64) == (64, 102) and OP.center_crop_origin(64, 102, 64) == (0, 19)
    assert OP.center_crop_origin(65, 64, 64) == (0, 0)          # round-half-even of 0.5


def test_oracle_matches_pillow_live():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.RandomState(5)
    for h, w in [(218, 178), (90, 121), (64, 200)]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        nh, nw = OP.resized_size(h, w, 64)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(OP.resize_bilinear_u8(img, nw, nh), ref)


@pytest.mark.parametrize('in_size,out_size', [(178, 64), (218, 78), (47, 97), (64, 64), (500, 64), (65, 64)])
def test_host_coefficients_match_oracle(in_size, out_size):
    kk, bounds, ks = PP._axis_tables(in_size, out_size)
    kk_o, bounds_o, ks_o = OP.resample_coeffs(in_size, out_size)
    assert ks == ks_o and np.array_equal(bounds, bounds_o) and np.array_equal(kk, kk_o)
    assert (kk.sum(axis=1) - (1 << OP.PRECISION_BITS)).__abs__().max() <= ks      # weights sum to one

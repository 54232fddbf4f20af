"""CPU restatement (numpy, integer arithmetic) of the image transform the reference's CelebA loaders
apply -- ``transforms.Compose([Resize(64), CenterCrop(64), ToTensor()])``, celeba/train.py:146-148,
celeba19/train.py:200-202 -- and of ``ToTensor`` alone (mnist/train.py:160,164).

TEST INFRASTRUCTURE: imported only by tests/ (checker for the HIP input pipeline).

The arithmetic lives in third-party code absent from /root/reference: torchvision (unpinned,
README.md:13) delegating to Pillow's ``Image.resize(..., BILINEAR)``.  Restated from Pillow's
published algorithm (src/libImaging/Resample.c: separable two-pass convolution, triangle filter whose
support scales with the reduction factor, coefficients normalised in double and rounded to 22-bit
fixed point, uint8 intermediate between the passes) and from torchvision's size / crop rules
(functional._compute_resized_output_size, functional.center_crop).  Pinned bit-exactly against
Pillow 12.2 itself: tests/golden/preprocess.npz (tests/golden/make_preprocess_golden.py) and, where
Pillow is importable, live in tests/test_preprocess_cpu.py.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """Bilinear (triangle) coefficients of one axis: (kk int32 [out, ksize], bounds int32 [out, 2] =
    (first tap, number of taps), ksize) -- precompute_coeffs + normalize_coeffs_8bpc."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * inv)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


def _resample_rows(img, out_size):
    """Resample axis 1 of a uint8 [H, W, C] image."""
    H, W, C = img.shape
    kk, bounds, _ = resample_coeffs(W, out_size)
    out = np.zeros((H, out_size, C), dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        acc += (img[:, xmin:xmin + xmax, :].astype(np.int64) * kk[xx, :xmax].astype(np.int64)[None, :, None]).sum(1)
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, out_w, out_h):
    """Pillow's Image.resize((out_w, out_h), BILINEAR) on a uint8 [H, W, C] array: horizontal pass,
    uint8 intermediate, vertical pass."""
    x = _resample_rows(img, out_w)
    return _resample_rows(x.transpose(1, 0, 2), out_h).transpose(1, 0, 2)


def resized_size(h, w, size):
    """torchvision Resize(int): the shorter side becomes ``size``, the longer int(size * long / short)."""
    if w <= h:
        return int(size * h / w), size          # (new_h, new_w)
    return size, int(size * w / h)


def center_crop_origin(h, w, size):
    """torchvision CenterCrop(size): (top, left) = round half to even of half the slack."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def to_tensor(u8_hwc):
    """torchvision ToTensor: uint8 [H, W, C] -> float32 [C, H, W] / 255."""
    return (u8_hwc.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


def resize_center_crop_to_tensor(u8_hwc, size=64):
    """Compose([Resize(size), CenterCrop(size), ToTensor()]) on one uint8 [H, W, 3] image."""
    h, w, _ = u8_hwc.shape
    nh, nw = resized_size(h, w, size)
    r = resize_bilinear_u8(u8_hwc, nw, nh)
    top, left = center_crop_origin(nh, nw, size)
    return to_tensor(r[top:top + size, left:left + size, :])

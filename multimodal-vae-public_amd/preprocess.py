"""On-device input pipeline: the per-image transforms of the reference's loaders as batch launches.

    ToTensor                      (mnist/train.py:160,164)        -> ``to_tensor(u8)``
    Compose([Resize(64), CenterCrop(64), ToTensor()])
                                  (celeba/train.py:146-148)       -> ``ResizeCenterCropToTensor(64)(u8_nhwc)``

Input: raw ``uint8`` images already in HBM (``[B, H, W]`` / ``[B, H, W, 3]``, the decoded JPEG / IDX
bytes); output: the ``float32`` NCHW batch the MVAE step consumes.  Resize is byte-exact with Pillow's
BILINEAR (what torchvision calls); the coefficient tables depend only on the image size and are built
once per size on the host by the library (``mvae_resample_coeffs``) and cached on the device.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .kernels import _need_gpu, _ptr, _stream, check


def to_tensor(u8):
    """uint8 [B, H, W] or [B, C, H, W] (cuda) -> float32 / 255; a missing channel axis is added."""
    _need_gpu(u8)
    if u8.dtype != torch.uint8:
        raise TypeError('to_tensor expects uint8, got %s' % u8.dtype)
    u8 = u8.contiguous()
    out = torch.empty(u8.shape, dtype=torch.float32, device=u8.device)
    check(_lib.lib().mvae_u8_to_f32(_ptr(u8), _ptr(out), u8.numel(), _stream()), 'mvae_u8_to_f32')
    return out.unsqueeze(1) if out.dim() == 3 else out


def resized_size(h, w, size):
    """torchvision Resize(int): shorter side -> size, longer -> int(size * long / short)."""
    return (int(size * h / w), size) if w <= h else (size, int(size * w / h))


def center_crop_origin(h, w, size):
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def _axis_tables(in_size, out_size):
    lib = _lib.lib()
    ks = lib.mvae_resample_ksize(in_size, out_size)
    check(min(ks, 0), 'mvae_resample_ksize')
    kk = np.zeros((out_size, ks), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    rc = lib.mvae_resample_coeffs(in_size, out_size, kk.ctypes.data_as(ctypes.c_void_p),
                                  bounds.ctypes.data_as(ctypes.c_void_p))
    check(min(rc, 0), 'mvae_resample_coeffs')
    return kk, bounds, ks


class ResizeCenterCropToTensor(object):
    """``transforms.Compose([Resize(size), CenterCrop(size), ToTensor()])`` for a uint8 NHWC batch."""

    def __init__(self, size=64):
        self.size = int(size)
        self._tables = {}

    def _plan(self, h, w, device):
        key = (h, w, device)
        if key not in self._tables:
            S = self.size
            nh, nw = resized_size(h, w, S)
            if nh < S or nw < S:
                raise ValueError('image %dx%d is smaller than the crop after Resize(%d)' % (h, w, S))
            top, left = center_crop_origin(nh, nw, S)
            kx, bx, ksx = _axis_tables(w, nw)
            ky, by, ksy = _axis_tables(h, nh)
            rows = by[top:top + S]
            y0, y1 = int(rows[:, 0].min()), int((rows[:, 0] + rows[:, 1]).max())
            dev = [torch.from_numpy(t).to(device) for t in (kx, bx, ky, by)]
            self._tables[key] = (nh, nw, top, left, ksx, ksy, y0, y1, dev)
        return self._tables[key]

    def __call__(self, u8_nhwc):
        _need_gpu(u8_nhwc)
        if u8_nhwc.dtype != torch.uint8 or u8_nhwc.dim() != 4 or u8_nhwc.shape[3] != 3:
            raise TypeError('expected a uint8 [B, H, W, 3] batch')
        x = u8_nhwc.contiguous()
        B, H, W, _ = x.shape
        nh, nw, top, left, ksx, ksy, y0, y1, (kx, bx, ky, by) = self._plan(H, W, x.device)
        S = self.size
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=x.device)
        check(_lib.lib().mvae_resize_crop_u8_to_f32(_ptr(x), _ptr(out), B, H, W, nh, nw, S, top, left, _ptr(kx),
                                                    _ptr(bx), ksx, _ptr(ky), _ptr(by), ksy, y0, y1, _stream()),
              'mvae_resize_crop_u8_to_f32')
        return out

#!/usr/bin/env python
"""Golden vectors for the input pipeline: what Pillow (the library torchvision's Resize / CenterCrop /
ToTensor delegate to for the reference's CelebA loaders, celeba/train.py:146-148) produces on seeded
random uint8 images.  Run in the build container (Pillow 12.2 is installed there):

    python tests/golden/make_preprocess_golden.py      # writes tests/golden/preprocess.npz
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import preprocess as OP  # noqa: E402

SHAPES = [(218, 178), (100, 160), (64, 64), (300, 65), (70, 130), (31, 47), (65, 64)]   # CelebA's is the first


def pil_pipeline(img, size=64):
    h, w, _ = img.shape
    nh, nw = (int(size * h / w), size) if w <= h else (size, int(size * w / h))
    r = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    return r, (r[top:top + size, left:left + size].astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


def main():
    rng = np.random.RandomState(20260927)
    out = {}
    for k, (h, w) in enumerate(SHAPES):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        if k == 1:
            img[:, :w // 2] = 255; img[:h // 2, w // 2:] = 0          # hard edges: saturation and rounding
        resized, final = pil_pipeline(img)
        nh, nw = OP.resized_size(h, w, 64)
        assert np.array_equal(OP.resize_bilinear_u8(img, nw, nh), resized), 'oracle != Pillow at %dx%d' % (h, w)
        assert np.array_equal(OP.resize_center_crop_to_tensor(img), final)
        out['image_%d' % k] = img
        out['resized_%d' % k] = resized
        out['final_%d' % k] = final
    np.savez_compressed(os.path.join(HERE, 'preprocess.npz'), **out)
    print('wrote preprocess.npz: %d cases, Pillow %s' % (len(SHAPES), Image.__version__))


if __name__ == '__main__':
    main()

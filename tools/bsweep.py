#!/usr/bin/env python
"""Batch sweep of the two 5 x 5 conv-forward-form launches (dec1 data gradient, enc4 forward): is a round batch a slow point?
    python tools/bsweep.py            (tuning library; hot re-issue in a hipGraph, tools/gemm_bench.timeit)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import torch  # noqa: E402

import gemm_bench as gb  # noqa: E402
import patch_bench as pb  # noqa: E402


def main():
    print('%-44s %8s %8s %8s' % ('op', 'GFLOP', 'TFLOP/s', 'us'))
    for B in (448, 480, 496, 504, 508, 509, 510, 511, 512, 513, 514, 516, 520, 528, 544, 576, 640):
        name, fl, fn, outs = pb.convT_dgrad(B, 256, 5, 128, 1, 0, 'dec1 256->128 5x5 s1 B%d' % B)
        fn(); torch.cuda.synchronize()
        ms = gb.timeit(fn, launches=10, replays=3)
        print('%-44s %8.2f %8.1f %8.1f' % (name, fl / 1e9, fl / (ms * 1e-3) / 1e12, ms * 1e3))
    for B in (224, 240, 248, 250, 252, 254, 255, 256, 257, 258, 260, 264, 272, 288, 320):
        name, fl, fn, outs = pb.conv_fwd(B, 128, 8, 256, 1, 0, 'enc4 128->256 8x8 s1 B%d' % B)
        fn(); torch.cuda.synchronize()
        ms = gb.timeit(fn, launches=10, replays=3)
        print('%-44s %8.2f %8.1f %8.1f' % (name, fl / 1e9, fl / (ms * 1e-3) / 1e12, ms * 1e3))


if __name__ == '__main__':
    main()

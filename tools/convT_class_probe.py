#!/usr/bin/env python
"""Transposed-conv forward launches of the parity-class kernel family, one at a time: us per launch and TFLOP/s.

    MVAE_HIP_LIB=.../libmvae_hip_tuning_<variant>.so python tools/convT_class_probe.py [tag]

Each launch is issued 3 times untimed, then 20 times between two events (operands as left by the previous launch: the
inputs of these shapes -- 51 / 151 / 34 / 67 MB -- do not stay in the 32 MB of L2), and prints the output's sum and sum of
squares in float64: variants that only reorder blocks or change the stores' cache policy must print the same digits.  Shapes: tools/traffic_probe.py CONV_CASES.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

CASES = [('convT2d_fwd', '2048x64x14x14'), ('convT2d_fwd', '4608x64x16x16'), ('convT2d_fwd', '512x32x32x32'),
         ('convT2d_fwd', '512x64x16x16')]


def main(tag):
    import torch
    import mvae_amd  # noqa: F401
    from mvae_amd import kernels as K
    import traffic_probe as T
    T.CONV_CASES[('convT2d_fwd', '512x64x16x16')] = ('convT', 512, 128, 8, 64, 2, 1)      # CelebA dec2
    for name, key in CASES:
        kind, B, Cin, H, Cout, s, p = T.CONV_CASES[(name, key)]
        OH = (H - 1) * s - 2 * p + 4
        taps = 4 if s == 2 else 16
        flops = 2.0 * B * Cout * OH * OH * Cin * taps
        torch.manual_seed(1234)
        x = torch.randn(B, Cin, H, H, device='cuda')
        w = torch.randn(Cin, Cout, 4, 4, device='cuda')
        y = torch.empty(B, Cout, OH, OH, device='cuda')
        fn = lambda: K.convT2d_fwd(x, w, y, None, s, p)  # noqa: E731
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        yd = y.double()      # same seed, same arithmetic per element: the sums must agree to the last digit across variants
        print('%-6s %s %-16s %8.1f us  %6.1f TFLOP/s  frac %.3f   sum %.10e  sumsq %.10e' % (
            tag, name, key, us, flops / us / 1e6, flops / us / 1e6 / 157.3, yd.sum().item(), (yd * yd).sum().item()))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'base')

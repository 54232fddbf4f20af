#!/usr/bin/env python
"""Micro-benchmark of the GEMM-shaped launches at the shapes of the bench workloads, for the
automatic tile/split plan and for forced tilings (mvae_debug_set_tiling).  Tuning aid:

    python tools/gemm_bench.py            # celeba B=256 + mnist B=512 shapes
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

os.environ.setdefault('MVAE_HIP_LIB', os.path.join(ROOT, 'multimodal-vae-public_amd', 'libmvae_hip_tuning.so'))

import torch  # noqa: E402

import mvae_amd  # noqa: E402
from mvae_amd import _lib, kernels as K  # noqa: E402

DEV = 'cuda'
PEAK = 157.3


def timeit(fn, launches=20, replays=5):
    """ms per call with the queue kept full: 20 calls captured into a hipGraph on a private stream,
    replayed between two HIP events (eager back-to-back launches are host-bound under ~15 us)."""
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(launches):
            fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        g.replay()
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
    st.synchronize()
    return e0.elapsed_time(e1) / (launches * replays)


def r(*shape):
    return torch.randn(*shape, device=DEV)


def conv_cases(B, Cin, H, Cout, s, p, tag):
    OH = (H + 2 * p - 4) // s + 1
    x, w = r(B, Cin, H, H), r(Cout, Cin, 4, 4)
    y, dy, dx, dw = torch.empty(B, Cout, OH, OH, device=DEV), r(B, Cout, OH, OH), torch.empty_like(x), torch.empty_like(w)
    fl = 2.0 * B * Cout * OH * OH * Cin * 16
    return [('%s conv fwd' % tag, fl, lambda: K.conv2d_fwd(x, w, y, None, s, p)),
            ('%s conv dgrad' % tag, fl, lambda: K.conv2d_dgrad(dy, w, dx, None, s, p)),
            ('%s conv wgrad' % tag, fl, lambda: K.conv2d_wgrad(dy, x, dw, s, p))]


def convT_cases(B, Cin, H, Cout, s, p, tag):
    OH = (H - 1) * s - 2 * p + 4
    x, w = r(B, Cin, H, H), r(Cin, Cout, 4, 4)
    y, dy, dx, dw = torch.empty(B, Cout, OH, OH, device=DEV), r(B, Cout, OH, OH), torch.empty_like(x), torch.empty_like(w)
    fl = 2.0 * B * Cin * H * H * Cout * 16
    return [('%s convT fwd' % tag, fl, lambda: K.convT2d_fwd(x, w, y, None, s, p)),
            ('%s convT dgrad' % tag, fl, lambda: K.convT2d_dgrad(dy, w, dx, None, s, p)),
            ('%s convT wgrad' % tag, fl, lambda: K.convT2d_wgrad(dy, x, dw, s, p))]


def lin_cases(M, N, Kd, tag):
    x, w, b = r(M, Kd), r(N, Kd), r(N)
    pre, act, dy, dx, dw, db = (torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV), r(M, N),
                                torch.empty(M, Kd, device=DEV), torch.empty(N, Kd, device=DEV),
                                torch.empty(N, device=DEV))
    fl = 2.0 * M * N * Kd
    return [('%s lin fwd' % tag, fl, lambda: K.linear_fwd(x, w, b, pre, act)),
            ('%s lin dgrad' % tag, fl, lambda: K.linear_dgrad(dy, w, dx)),
            ('%s lin wgrad' % tag, fl, lambda: K.linear_wgrad(dy, x, dw, db))]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='all', choices=['all', 'lin', 'conv', 'knockout', 'dec19', 'biglin'])
    ap.add_argument('--auto-only', action='store_true', help='time only the automatic plan')
    args = ap.parse_args()
    conv, lin = [], []
    B = 256
    if args.cases == 'dec19':
        # the image decoder's forward over the rows of celeba19's 21 terms (the three largest kernels of that step)
        rows = 21 * B
        for name, fl, fn in (convT_cases(rows, 256, 5, 128, 1, 0, 'c19 dec1 256->128 5x5 s1')[:1] +
                             convT_cases(rows, 128, 8, 64, 2, 1, 'c19 dec2 128->64 8x8')[:1] +
                             convT_cases(rows, 64, 16, 32, 2, 1, 'c19 dec3 64->32 16x16')[:1] +
                             convT_cases(rows, 32, 32, 3, 2, 1, 'c19 dec4 32->3 32x32')[:1]):
            ms = timeit(fn, launches=4, replays=3)
            print('%-34s %8.2f GFLOP %8.1f TFLOP/s %9.1f us' % (name, fl / 1e9, fl / (ms * 1e-3) / 1e12, ms * 1e3))
        return
    if args.cases == 'biglin':
        # the 6272- / 6400-wide Linear layers (FashionMNIST B = 1024: two decoder terms = 2048 rows; CelebA B = 256)
        # under the wide tilings: is 64x64 still the right tile for a GEMM with thousands of tiles?
        big = (lin_cases(2048, 6272, 512, 'fmnist 512->6272 M2048') + lin_cases(1024, 512, 6272, 'fmnist 6272->512 M1024') +
               lin_cases(256, 512, 6400, 'celeba 6400->512 M256') + lin_cases(768, 6400, 100, 'celeba 100->6400 M768'))
        lib = _lib.lib()
        cfgs = [('auto', (0, 0, 0)), ('64x64', (1, 1, 0)), ('64x128', (1, 2, 0)), ('128x64', (2, 1, 0)), ('128x128', (2, 2, 0)),
                ('64x128 s1', (1, 2, 1)), ('128x128 s1', (2, 2, 1))]
        print('%-34s %8s | ' % ('op', 'GFLOP') + ' '.join('%11s' % c[0] for c in cfgs) + '   (TFLOP/s)')
        for name, fl, fn in big:
            row = []
            for cname, (wm, wn, sp) in cfgs:
                lib.mvae_debug_set_tiling(wm, wn, sp)
                lib.mvae_debug_set_small(1 if wm else 0, 0)
                try:
                    row.append(fl / (timeit(fn, launches=8, replays=3) * 1e-3) / 1e12)
                except RuntimeError:
                    row.append(float('nan'))
            lib.mvae_debug_set_tiling(0, 0, 0); lib.mvae_debug_set_small(0, 0)
            print('%-34s %8.2f | ' % (name, fl / 1e9) + ' '.join('%11.1f' % v for v in row))
        return
    conv += conv_cases(B, 3, 64, 32, 2, 1, 'enc1 3->32 64x64')
    conv += conv_cases(B, 32, 32, 64, 2, 1, 'enc2 32->64 32x32')
    conv += conv_cases(B, 64, 16, 128, 2, 1, 'enc3 64->128 16x16')
    conv += conv_cases(B, 128, 8, 256, 1, 0, 'enc4 128->256 8x8 s1')
    conv += convT_cases(2 * B, 256, 5, 128, 1, 0, 'dec1 256->128 5x5 s1')
    conv += convT_cases(2 * B, 128, 8, 64, 2, 1, 'dec2 128->64 8x8')
    conv += convT_cases(2 * B, 64, 16, 32, 2, 1, 'dec3 64->32 16x16')
    conv += convT_cases(2 * B, 32, 32, 3, 2, 1, 'dec4 32->3 32x32')
    lin += lin_cases(B, 512, 6400, 'celeba 6400->512 M256')
    lin += lin_cases(2 * B, 6400, 100, 'celeba 100->6400 M512')
    lin += lin_cases(3 * B, 512, 512, 'celeba attr 512->512 M768')
    lin += lin_cases(1024, 512, 512, 'mnist 512->512 M1024')
    lin += lin_cases(512, 512, 512, 'mnist 512->512 M512')
    lin += lin_cases(512, 512, 784, 'mnist 784->512 M512')
    lin += lin_cases(1024, 784, 512, 'mnist 512->784 M1024')
    lin += lin_cases(1024, 512, 64, 'mnist 64->512 M1024')
    lin += lin_cases(512, 128, 512, 'mnist heads 512->128 M512')
    lin += lin_cases(1024, 6272, 512, 'fmnist 512->6272 M1024')
    lin += lin_cases(1024, 512, 6272, 'fmnist 6272->512 M1024')
    lib = _lib.lib()
    # (label, (wm, wn, splits), kwaves, (small_off, small_waves))
    conv_cfg = [('auto', (0, 0, 0), 0, (0, 0)), ('64x64 kw1', (1, 1, 0), 1, (0, 0)), ('128x128', (2, 2, 0), 1, (0, 0)),
                ('64x128', (1, 2, 0), 1, (0, 0)), ('128x64', (2, 1, 0), 1, (0, 0))]
    lin_cfg = [('auto', (0, 0, 0), 0, (0, 0)), ('r1 plan', (0, 0, 0), 0, (1, 0)), ('small w4', (0, 0, 0), 0, (0, 4)),
               ('small w8', (0, 0, 0), 0, (0, 8)), ('64x64 s1kw1', (1, 1, 1), 1, (1, 0)), ('s1 kw4', (1, 1, 1), 4, (1, 0))]
    if args.cases == 'knockout':
        # where the time of the direct Linear weight-gradient launches goes: loads only / MFMAs only
        print('%-34s %10s %10s %10s   (us per launch)' % ('op', 'kernel', 'loads only', 'mfma only'))
        for name, fl, fn in lin:
            if 'wgrad' not in name:
                continue
            row = []
            for mode in (0, 1, 2):
                lib.mvae_debug_set_knockout(mode)
                row.append(timeit(fn) * 1e3)
            lib.mvae_debug_set_knockout(0)
            print('%-34s %10.1f %10.1f %10.1f' % tuple([name] + row))
        return
    if args.auto_only:
        conv_cfg, lin_cfg = conv_cfg[:1], lin_cfg[:1]
    tot = {}
    for title, cases, configs in (('conv', conv, conv_cfg), ('lin', lin, lin_cfg)):
        if args.cases not in ('all', title):
            continue
        print('%-34s %8s | ' % ('op', 'GFLOP') + ' '.join('%11s' % c[0] for c in configs) + '   (TFLOP/s; us for auto)')
        for name, fl, fn in cases:
            row = []
            for cname, (wm, wn, sp), kw, (soff, sw) in configs:
                lib.mvae_debug_set_tiling(wm, wn, sp)
                lib.mvae_debug_set_kwaves(kw)
                lib.mvae_debug_set_small(soff, sw)
                try:
                    ms = timeit(fn)
                    row.append(fl / (ms * 1e-3) / 1e12)
                    tot[(title, cname)] = tot.get((title, cname), 0.0) + ms
                    if cname == 'auto':
                        t_auto = ms
                except RuntimeError:
                    row.append(float('nan'))
            lib.mvae_debug_set_tiling(0, 0, 0)
            lib.mvae_debug_set_kwaves(0)
            lib.mvae_debug_set_small(0, 0)
            print('%-34s %8.2f | ' % (name, fl / 1e9) + ' '.join('%11.1f' % v for v in row) + '   %8.1f us' % (t_auto * 1e3))
        print('sum of times (ms): ' + '  '.join('%s %.3f' % (c[0], tot.get((title, c[0]), float('nan'))) for c in configs))


if __name__ == '__main__':
    main()

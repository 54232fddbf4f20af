#!/usr/bin/env python
"""Version-2 GEMM core (csrc/gemm2.h) against the round-1-5 kernels at the shapes of the bench workloads: outputs compared
element by element, launches timed hot inside a hipGraph (tools/gemm_bench.timeit).  Uses the tuning library, whose
MVAE_G2_OFF / MVAE_G2_FORCE=wm,wn,occ environment switches pick the kernel per call.

    python tools/g2_bench.py [--cases lin|conv|all] [--sweep]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import torch  # noqa: E402

import gemm_bench as gb  # noqa: E402  (sets MVAE_HIP_LIB to the tuning build)
from mvae_amd import kernels as K  # noqa: E402

PEAK = 157.3


def lin_cases(M, N, Kd, tag):
    r = gb.r
    x, w, b = r(M, Kd), r(N, Kd), r(N)
    pre, act, dy = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda'), r(M, N)
    dx, dw, db = torch.empty(M, Kd, device='cuda'), torch.empty(N, Kd, device='cuda'), torch.empty(N, device='cuda')
    fl = 2.0 * M * N * Kd
    return [('%s fwd' % tag, fl, lambda: K.linear_fwd(x, w, b, pre, act), lambda: (pre, act)),
            ('%s fwd (pre only)' % tag, fl, lambda: K.linear_fwd(x, w, b, pre, None), lambda: (pre,)),
            ('%s dgrad' % tag, fl, lambda: K.linear_dgrad(dy, w, dx, pre_in=x), lambda: (dx,)),
            ('%s wgrad' % tag, fl, lambda: K.linear_wgrad(dy, x, dw, db), lambda: (dw, db))]


def glin_cases(G, M, N, Kd, tag):
    r = gb.r
    x, w, b = r(G, M, Kd), r(G, N, Kd), r(G, N)
    pre, act, dy = torch.empty(G, M, N, device='cuda'), torch.empty(G, M, N, device='cuda'), r(G, M, N)
    dx, dw, db = torch.empty(G, M, Kd, device='cuda'), torch.empty(G, N, Kd, device='cuda'), torch.empty(G, N, device='cuda')
    fl = 2.0 * G * M * N * Kd
    return [('%s fwd' % tag, fl, lambda: K.linear_fwd_grouped(x, w[0], N * Kd, b[0], N, pre, act), lambda: (pre, act)),
            ('%s dgrad' % tag, fl, lambda: K.linear_dgrad_grouped(dy, w[0], N * Kd, dx, pre_in=x), lambda: (dx,)),
            ('%s wgrad' % tag, fl, lambda: K.linear_wgrad_grouped(dy, x, dw[0], N * Kd, db[0], N), lambda: (dw, db))]


def conv_cases(B, Cin, H, Cout, s, p, tag):
    r = gb.r
    OH = (H + 2 * p - 4) // s + 1
    x, w = r(B, Cin, H, H), r(Cout, Cin, 4, 4)
    y, a, dy = torch.empty(B, Cout, OH, OH, device='cuda'), torch.empty(B, Cout, OH, OH, device='cuda'), r(B, Cout, OH, OH)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    fl = 2.0 * B * Cout * OH * OH * Cin * 16
    return [('%s conv fwd' % tag, fl, lambda: K.conv2d_fwd(x, w, y, a, s, p), lambda: (y, a)),
            ('%s conv dgrad' % tag, fl, lambda: K.conv2d_dgrad(dy, w, dx, x, s, p), lambda: (dx,)),
            ('%s conv wgrad' % tag, fl, lambda: K.conv2d_wgrad(dy, x, dw, s, p), lambda: (dw,))]


def convT_cases(B, Cin, H, Cout, s, p, tag):
    r = gb.r
    OH = (H - 1) * s - 2 * p + 4
    x, w = r(B, Cin, H, H), r(Cin, Cout, 4, 4)
    y, a, dy = torch.empty(B, Cout, OH, OH, device='cuda'), torch.empty(B, Cout, OH, OH, device='cuda'), r(B, Cout, OH, OH)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    fl = 2.0 * B * Cin * H * H * Cout * 16
    return [('%s convT fwd' % tag, fl, lambda: K.convT2d_fwd(x, w, y, a, s, p), lambda: (y, a)),
            ('%s convT dgrad' % tag, fl, lambda: K.convT2d_dgrad(dy, w, dx, x, s, p), lambda: (dx,)),
            ('%s convT wgrad' % tag, fl, lambda: K.convT2d_wgrad(dy, x, dw, s, p), lambda: (dw,))]


def set_mode(mode):
    os.environ.pop('MVAE_G2_OFF', None)
    os.environ.pop('MVAE_G2_FORCE', None)
    if mode == 'v1':
        os.environ['MVAE_G2_OFF'] = '1'
    elif mode != 'auto':
        os.environ['MVAE_G2_FORCE'] = mode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='lin')
    ap.add_argument('--sweep', action='store_true', help='forced tile shapes / occupancies besides the automatic plan')
    args = ap.parse_args()
    cases = []
    if args.cases in ('lin', 'all'):
        cases += lin_cases(2048, 6272, 512, 'fmnist 512->6272 M2048')
        cases += lin_cases(1024, 512, 6272, 'fmnist 6272->512 M1024')
        cases += lin_cases(2048, 512, 512, 'fmnist 512->512 M2048')
        cases += lin_cases(256, 512, 6400, 'celeba 6400->512 M256')
        cases += lin_cases(4608, 6400, 100, 'celeba19 100->6400 M4608')
        cases += lin_cases(1000, 520, 100, 'ragged 100->520 M1000')
    if args.cases in ('glin', 'all'):
        cases += glin_cases(18, 768, 512, 512, 'c19 G18 512->512 M768')
        cases += glin_cases(18, 256, 512, 512, 'c19 G18 512->512 M256')
        cases += glin_cases(18, 768, 512, 100, 'c19 G18 100->512 M768')
        cases += glin_cases(5, 100, 52, 36, 'ragged G5 36->52 M100')
    if args.cases in ('conv', 'all'):
        B = 256
        cases += conv_cases(B, 32, 32, 64, 2, 1, 'enc2 32->64 32x32')
        cases += conv_cases(B, 64, 16, 128, 2, 1, 'enc3 64->128 16x16')
        cases += conv_cases(B, 128, 8, 256, 1, 0, 'enc4 128->256 8x8 s1')
        cases += convT_cases(2 * B, 256, 5, 128, 1, 0, 'dec1 256->128 5x5 s1')
        cases += convT_cases(2 * B, 128, 8, 64, 2, 1, 'dec2 128->64 8x8')
        cases += convT_cases(2 * B, 64, 16, 32, 2, 1, 'dec3 64->32 16x16')
        cases += conv_cases(1024, 64, 14, 128, 2, 1, 'fm enc2 64->128 14x14')
        cases += convT_cases(2048, 128, 7, 64, 2, 1, 'fm dec2 128->64 7x7')
    modes = ['v1', 'auto']
    if args.sweep:
        modes += ['2,2,2', '2,2,-1', '2,1,2', '2,1,-1', '1,2,2', '1,2,-1', '1,1,4', '1,1,2', '1,1,-1']
    print('%-36s %7s | ' % ('op', 'GFLOP') + ' '.join('%9s' % m for m in modes) + '   (TFLOP/s)   | max rel diff of auto vs v1')
    for name, fl, fn, outs in cases:
        row, ref, worst = [], None, 0.0
        for m in modes:
            set_mode(m)
            for o in outs():
                o.fill_(float('nan'))
            try:
                fn()
                torch.cuda.synchronize()
                got = [o.clone() for o in outs()]
                if m == 'v1':
                    ref = got
                else:
                    for g, rf in zip(got, ref):
                        d = ((g - rf).abs().max() / rf.abs().max().clamp_min(1e-30)).item()
                        if d != d:
                            d = float('inf')
                        worst = max(worst, d)
                row.append(fl / (gb.timeit(fn, launches=10, replays=3) * 1e-3) / 1e12)
            except RuntimeError as ex:
                row.append(float('nan'))
                print('   (%s: %s)' % (m, ex))
        set_mode('auto')
        print('%-36s %7.2f | ' % (name, fl / 1e9) + ' '.join('%9.1f' % v for v in row) + '   | %.2e' % worst)


if __name__ == '__main__':
    main()

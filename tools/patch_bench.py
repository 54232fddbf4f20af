#!/usr/bin/env python
"""The LDS-patch transposed-conv kernels (csrc/convt_patch.h) against the class-by-class gather launches of gemm_core.h at
the shapes of the bench workloads: outputs compared element by element, launches timed hot inside a hipGraph
(tools/gemm_bench.timeit).  Uses the tuning library (MVAE_PATCH_OFF picks the kernel per call).

    python tools/patch_bench.py [lib-variant ...]     (extra columns: MVAE_HIP_LIB variants cannot switch in-process -- one process per library)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import torch  # noqa: E402

import gemm_bench as gb  # noqa: E402  (sets MVAE_HIP_LIB to the tuning build unless it is set)
from mvae_amd import kernels as K  # noqa: E402


def convT_fwd(B, Cin, H, Cout, tag, stats=False):
    r = gb.r
    OH = 2 * H
    x, w = r(B, Cin, H, H), r(Cin, Cout, 4, 4)
    y, a = torch.empty(B, Cout, OH, OH, device='cuda'), torch.empty(B, Cout, OH, OH, device='cuda')
    fl = 2.0 * B * Cin * H * H * Cout * 16
    return ('%s convT fwd' % tag, fl, lambda: K.convT2d_fwd(x, w, y, a, 2, 1), lambda: (y, a))


def convT_stats(B, Cin, H, Cout, tag):
    r = gb.r
    x, w = r(B, Cin, H, H), r(Cin, Cout, 4, 4)
    fl = 2.0 * B * Cin * H * H * Cout * 16
    box = {}

    def fn():
        box['part'] = K.convT2d_fwd_stats(x, w, 2, 1)
    fn()
    return ('%s convT fwd stats' % tag, fl, fn, lambda: (box['part'],))


def conv_dgrad(B, Cin, H, Cout, tag):
    r = gb.r
    OH = H // 2
    x, w, dy = r(B, Cin, H, H), r(Cout, Cin, 4, 4), r(B, Cout, OH, OH)
    dx = torch.empty_like(x)
    fl = 2.0 * B * Cout * OH * OH * Cin * 16
    return ('%s conv dgrad' % tag, fl, lambda: K.conv2d_dgrad(dy, w, dx, x, 2, 1), lambda: (dx,))


def wgrad(B, Cin, H, Cout, tag, transposed):
    """Conv2d(Cin, Cout) on H x H maps / ConvTranspose2d(Cin, Cout) on H x H maps (stride 2, pad 1)."""
    r = gb.r
    if transposed:
        x, dy, dw = r(B, Cin, H, H), r(B, Cout, 2 * H, 2 * H), torch.empty(Cin, Cout, 4, 4, device='cuda')
        fl = 2.0 * B * Cin * H * H * Cout * 16
        return ('%s convT wgrad' % tag, fl, lambda: K.convT2d_wgrad(dy, x, dw, 2, 1), lambda: (dw,))
    x, dy, dw = r(B, Cin, H, H), r(B, Cout, H // 2, H // 2), torch.empty(Cout, Cin, 4, 4, device='cuda')
    fl = 2.0 * B * Cout * (H // 2) ** 2 * Cin * 16
    return ('%s conv wgrad' % tag, fl, lambda: K.conv2d_wgrad(dy, x, dw, 2, 1), lambda: (dw,))


def wgrad_s1(B, Cin, Cout, tag, transposed):
    """Conv2d(Cin, Cout, 4, 1, 0) on 8 x 8 maps / ConvTranspose2d(Cin, Cout, 4, 1, 0) on 5 x 5 maps."""
    r = gb.r
    if transposed:
        x, dy, dw = r(B, Cin, 5, 5), r(B, Cout, 8, 8), torch.empty(Cin, Cout, 4, 4, device='cuda')
        fl = 2.0 * B * Cin * 25 * Cout * 16
        return ('%s convT wgrad' % tag, fl, lambda: K.convT2d_wgrad(dy, x, dw, 1, 0), lambda: (dw,))
    x, dy, dw = r(B, Cin, 8, 8), r(B, Cout, 5, 5), torch.empty(Cout, Cin, 4, 4, device='cuda')
    fl = 2.0 * B * Cout * 25 * Cin * 16
    return ('%s conv wgrad' % tag, fl, lambda: K.conv2d_wgrad(dy, x, dw, 1, 0), lambda: (dw,))


def conv_fwd(B, Cin, H, Cout, s_, p_, tag):
    r = gb.r
    OH = (H + 2 * p_ - 4) // s_ + 1
    x, w = r(B, Cin, H, H), r(Cout, Cin, 4, 4)
    y, a = torch.empty(B, Cout, OH, OH, device='cuda'), torch.empty(B, Cout, OH, OH, device='cuda')
    fl = 2.0 * B * Cout * OH * OH * Cin * 16
    return ('%s conv fwd' % tag, fl, lambda: K.conv2d_fwd(x, w, y, a, s_, p_), lambda: (y, a))


def convT_dgrad(B, Cin, H, Cout, s_, p_, tag):
    """Data gradient of ConvTranspose2d(Cin, Cout) on H x H inputs: dy [B, Cout, OH, OH] -> dx [B, Cin, H, H] (x Swish'(pre))."""
    r = gb.r
    OH = (H - 1) * s_ - 2 * p_ + 4
    x, w, dy = r(B, Cin, H, H), r(Cin, Cout, 4, 4), r(B, Cout, OH, OH)
    dx = torch.empty_like(x)
    fl = 2.0 * B * Cin * H * H * Cout * 16
    return ('%s convT dgrad' % tag, fl, lambda: K.convT2d_dgrad(dy, w, dx, x, s_, p_), lambda: (dx,))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if len(sys.argv) > 2:
        os.environ['MVAE_WGRAD_PATCH_TARGET'] = sys.argv[2]
    switch = {'wgrad': 'MVAE_WGRAD_PATCH_OFF', 'fwd': 'MVAE_CONV_PATCH_OFF', 'tail': 'MVAE_TAIL_OFF'}.get(which, 'MVAE_PATCH_OFF')
    fcases = [conv_fwd(256, 32, 32, 64, 2, 1, 'enc2 32->64 32x32 B256'),
              conv_fwd(256, 64, 16, 128, 2, 1, 'enc3 64->128 16x16 B256'),
              conv_fwd(256, 128, 8, 256, 1, 0, 'enc4 128->256 8x8 s1 B256'),
              convT_dgrad(512, 256, 5, 128, 1, 0, 'dec1 256->128 5x5 s1 B512'),
              convT_dgrad(512, 128, 8, 64, 2, 1, 'dec2 128->64 8x8 B512'),
              convT_dgrad(512, 64, 16, 32, 2, 1, 'dec3 64->32 16x16 B512'),
              conv_fwd(1024, 64, 14, 128, 2, 1, 'fm enc2 64->128 14x14 B1024'),
              convT_dgrad(2048, 128, 7, 64, 2, 1, 'fm dec2 128->64 7x7 B2048'),
              conv_fwd(37, 64, 14, 128, 2, 1, 'ragged 64->128 14x14 B37')]
    tcases = [convT_dgrad(512, 256, 5, 128, 1, 0, 'dec1 256->128 5x5 s1 B512'),
              conv_fwd(256, 128, 8, 256, 1, 0, 'enc4 128->256 8x8 s1 B256'),
              conv_fwd(1024, 64, 14, 128, 2, 1, 'fm enc2 64->128 14x14 B1024'),
              convT_dgrad(2048, 128, 7, 64, 2, 1, 'fm dec2 128->64 7x7 B2048'),
              convT_dgrad(4608, 256, 5, 128, 1, 0, 'dec1 256->128 5x5 s1 B4608'),
              convT_dgrad(509, 256, 5, 128, 1, 0, 'ragged dec1 256->128 5x5 s1 B509'),
              conv_fwd(250, 128, 8, 256, 1, 0, 'ragged enc4 128->256 8x8 s1 B250')] if which == 'tail' else []
    wcases = [wgrad(256, 32, 32, 64, 'enc2 32->64 32x32 B256', False),
              wgrad(256, 64, 16, 128, 'enc3 64->128 16x16 B256', False),
              wgrad(512, 128, 8, 64, 'dec2 128->64 8x8 B512', True),
              wgrad(512, 64, 16, 32, 'dec3 64->32 16x16 B512', True),
              wgrad(4608, 128, 8, 64, 'dec2 128->64 8x8 B4608', True),
              wgrad(7, 64, 16, 128, 'ragged 64->128 16x16 B7', False),
              wgrad_s1(256, 128, 256, 'enc4 128->256 8x8 s1 B256', False),
              wgrad_s1(512, 256, 128, 'dec1 256->128 5x5 s1 B512', True),
              wgrad_s1(4608, 256, 128, 'dec1 256->128 5x5 s1 B4608', True),
              wgrad_s1(255, 128, 256, 'enc4 128->256 8x8 s1 B255', False)]
    cases = wcases if which == 'wgrad' else fcases if which == 'fwd' else tcases if which == 'tail' else [convT_fwd(2048, 128, 7, 64, 'fm dec2 128->64 7x7 B2048'),
             conv_dgrad(1024, 64, 14, 128, 'fm enc2 64->128 14x14 B1024'),
             convT_fwd(512, 128, 8, 64, 'dec2 128->64 8x8 B512'),
             convT_fwd(4608, 128, 8, 64, 'dec2 128->64 8x8 B4608'),
             conv_dgrad(256, 64, 16, 128, 'enc3 64->128 16x16 B256'),
             convT_fwd(512, 64, 16, 32, 'dec3 64->32 16x16 B512'),
             convT_fwd(256, 64, 16, 32, 'dec3 64->32 16x16 B256'),
             conv_dgrad(256, 32, 32, 64, 'enc2 32->64 32x32 B256'),
             convT_stats(4608, 64, 16, 32, 'dec3 64->32 16x16 B4608'),
             convT_stats(256, 64, 16, 32, 'dec3 64->32 16x16 B256')]
    print('%-40s %7s | %9s %9s  (TFLOP/s, us)  | max rel diff' % ('op', 'GFLOP', 'gather', 'patch'))
    for name, fl, fn, outs in cases:
        row, ref, worst = [], None, 0.0
        for off in (True, False):
            if off:
                os.environ[switch] = '1'
            else:
                os.environ.pop(switch, None)
            for o in outs():
                o.fill_(float('nan'))
            fn()
            torch.cuda.synchronize()
            got = [o.clone() for o in outs()]
            if off:
                ref = got
            else:
                for g, rf in zip(got, ref):
                    d = ((g - rf).abs().max() / rf.abs().max().clamp_min(1e-30)).item()
                    worst = max(worst, d if d == d else float('inf'))
            row.append(gb.timeit(fn, launches=10, replays=3))
        os.environ.pop(switch, None)
        print('%-40s %7.2f | ' % (name, fl / 1e9) + ' '.join('%6.1f %6.1f' % (fl / (ms * 1e-3) / 1e12, ms * 1e3) for ms in row) + '   | %.2e' % worst)


if __name__ == '__main__':
    main()

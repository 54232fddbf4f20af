#!/usr/bin/env python
"""convT_s1_kernel (the stride-1 5 x 5 -> 8 x 8 transposed conv as a dense GEMM + col2im) at the shapes of the bench workloads:
hot re-issue in a hipGraph (tools/gemm_bench.timeit), one process per library variant (MVAE_HIP_LIB), a checksum of the outputs so
that variants which must agree can be compared line by line.     python tools/s1_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import torch  # noqa: E402

import gemm_bench as gb  # noqa: E402
from mvae_amd import kernels as K  # noqa: E402


def main():
    torch.manual_seed(0)
    r = lambda *s: torch.randn(*s, device='cuda')  # noqa: E731
    print('# lib %s' % os.path.basename(os.environ.get('MVAE_HIP_LIB', '')))
    print('%-44s %8s %8s %8s   %s' % ('op', 'GFLOP', 'TFLOP/s', 'us', 'checksum (sum, sum |.|)'))
    rows = []
    for B in (512, 256, 4608, 509):
        x, w = r(B, 256, 5, 5), r(256, 128, 4, 4)
        y, a = torch.empty(B, 128, 8, 8, device='cuda'), torch.empty(B, 128, 8, 8, device='cuda')
        rows.append(('convT fwd 256->128 5x5->8x8 B%d' % B, 2.0 * B * 256 * 25 * 128 * 16, lambda x=x, w=w, y=y, a=a: K.convT2d_fwd(x, w, y, a, 1, 0), (y, a)))
    for B in (256, 250):
        x, w, dy = r(B, 128, 8, 8), r(256, 128, 4, 4), r(B, 256, 5, 5)
        dx = torch.empty_like(x)
        rows.append(('conv dgrad 128<-256 8x8<-5x5 B%d' % B, 2.0 * B * 256 * 25 * 128 * 16, lambda dy=dy, w=w, dx=dx, x=x: K.conv2d_dgrad(dy, w, dx, x, 1, 0), (dx,)))
    for rep in range(2):                    # the first pass of a process reads slow (clocks): print the second
        out = []
        for name, fl, fn, outs in rows:
            for o in outs:
                o.fill_(float('nan'))
            fn()
            torch.cuda.synchronize()
            cs = ' '.join('%.6e %.6e' % (o.double().sum().item(), o.double().abs().sum().item()) for o in outs)
            ms = gb.timeit(fn, launches=10, replays=3)
            out.append('%-44s %8.2f %8.1f %8.1f   %s' % (name, fl / 1e9, fl / (ms * 1e-3) / 1e12, ms * 1e3, cs))
    print('\n'.join(out))


if __name__ == '__main__':
    main()

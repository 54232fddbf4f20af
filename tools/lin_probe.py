#!/usr/bin/env python
"""Linear fwd / dgrad / wgrad of given (M, N, K) shapes against torch fp64 on the CPU (diagnostic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mvae_amd  # noqa: E402
from mvae_amd import kernels as K  # noqa: E402


def rel(a, b):
    b = b.double()
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    shapes = [tuple(int(v) for v in s.split(',')) for s in sys.argv[1:]]
    for M, N, Kd in shapes:
        g = torch.Generator().manual_seed(M + N + Kd)
        x, w, b = torch.randn(M, Kd, generator=g), torch.randn(N, Kd, generator=g) * Kd ** -0.5, torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g)
        xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
        pre = torch.empty(M, N, device='cuda'); act = torch.empty(M, N, device='cuda')
        K.linear_fwd(xd, wd, bd, pre, act)
        ref = x.double() @ w.double().t() + b.double()
        e_f = rel(pre, ref)
        dx = torch.empty(M, Kd, device='cuda')
        K.linear_dgrad(dyd, wd, dx)
        e_d = rel(dx, dy.double() @ w.double())
        base = torch.randn(M, Kd, generator=g)
        dx2 = base.cuda()
        K.linear_dgrad(dyd, wd, dx2, accumulate=True)
        e_da = rel(dx2, base.double() + dy.double() @ w.double())
        dw = torch.empty(N, Kd, device='cuda'); db = torch.empty(N, device='cuda')
        K.linear_wgrad(dyd, xd, dw, db)
        e_w = rel(dw, dy.double().t() @ x.double()); e_b = rel(db, dy.double().sum(0))
        flag = '<<<' if max(e_f, e_d, e_da, e_w, e_b) > 1e-4 else ''
        print('M%-5d N%-5d K%-5d fwd %.2e dgrad %.2e dgrad+acc %.2e wgrad %.2e bias %.2e %s' % (M, N, Kd, e_f, e_d, e_da, e_w, e_b, flag))


if __name__ == '__main__':
    main()

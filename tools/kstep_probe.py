#!/usr/bin/env python
"""Time of one unsplit 64x64-tile Linear forward vs reduction length: slope = time per k-step
(BK = 32), intercept = launch + prologue + epilogue.  Tuning aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import os as _os
_os.environ.setdefault("MVAE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multimodal-vae-public_amd", "libmvae_hip_tuning.so"))
import torch  # noqa: E402

import mvae_amd  # noqa: F401,E402
from mvae_amd import _lib, kernels as K  # noqa: E402
from gemm_bench import timeit  # noqa: E402


def graph_time(fn, n=20, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        g.replay(); e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
    st.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


lib = _lib.lib()
for M, N in ((1024, 512), (4096, 2048)):
    for kw in (1, 4):
        lib.mvae_debug_set_tiling(1, 1, 1); lib.mvae_debug_set_kwaves(kw)
        row = []
        for Kd in (32, 64, 128, 256, 512, 1024, 2048, 4096):
            x, w, b = torch.randn(M, Kd, device='cuda'), torch.randn(N, Kd, device='cuda'), torch.randn(N, device='cuda')
            pre, act = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
            us = graph_time(lambda: K.linear_fwd(x, w, b, pre, act))
            row.append('K%d: %.1fus' % (Kd, us))
        print('M%d N%d kw%d  ' % (M, N, kw) + '  '.join(row))
lib.mvae_debug_set_tiling(0, 0, 0); lib.mvae_debug_set_kwaves(0)

# fixed cost of a launch: one tile, one k-step; then the epilogue variants
print('--- floor')
for M, N, Kd, two in ((64, 64, 32, True), (64, 64, 32, False), (1024, 512, 32, True), (1024, 512, 32, False),
                      (1024, 512, 128, True), (1024, 512, 128, False)):
    lib.mvae_debug_set_tiling(1, 1, 1); lib.mvae_debug_set_kwaves(1)
    x, w, b = torch.randn(M, Kd, device='cuda'), torch.randn(N, Kd, device='cuda'), torch.randn(N, device='cuda')
    pre, act = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    us = graph_time(lambda: K.linear_fwd(x, w, b, pre, act if two else None))
    print('M%d N%d K%d %s: %.2f us' % (M, N, Kd, 'pre+act' if two else 'pre only', us))
lib.mvae_debug_set_tiling(0, 0, 0); lib.mvae_debug_set_kwaves(0)
z = torch.zeros(1 << 20, device='cuda')
print('fill 4 MB: %.2f us' % graph_time(lambda: K.fill_(z, 1.0)))
z2 = torch.zeros(256, device='cuda')
print('fill 1 KB: %.2f us' % graph_time(lambda: K.fill_(z2, 1.0)))

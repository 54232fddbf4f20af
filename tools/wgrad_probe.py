#!/usr/bin/env python
"""GPU box: time ONE batched Linear weight-gradient launch (mvae_linear_wgrad_batched) on MNIST's real batches, under the
library named by MVAE_HIP_LIB (tools/build_variants.sh):

    python tools/wgrad_probe.py [tag]

  image-dec   the image decoder's four layers, M = 1024 rows (64->512, 512->512, 512->512, 512->784): 1.96 GFLOP
  label-dec   the label decoder's four layers, M = 1024 (64->512, 512->512, 512->512, 512->10):       1.15 GFLOP
  img-enc     the image encoder's three layers, M = 512 (784->512, 512->512, 512->128 heads)
hot  = the launch re-issued 20x inside one hipGraph on the same operands (L2 / MALL hot), per-launch average;
cold = each launch behind a 512-MB fill that evicts L2 and the Infinity Cache, timed with an event pair around the launch
       alone (includes one launch boundary).
Prints one line per batch: us hot, us cold, TFLOP/s hot, fraction of the fp32 MFMA peak (157.3 TFLOP/s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mvae_amd  # noqa: E402,F401
from mvae_amd import kernels as K  # noqa: E402

PEAK = 157.3
DEV = 'cuda'


def batch(M, layers):
    items, flops = [], 0
    for (n_in, n_out) in layers:
        dy = torch.randn(M, n_out, device=DEV)
        x = torch.randn(M, n_in, device=DEV)
        dw = torch.empty(n_out, n_in, device=DEV)
        db = torch.empty(n_out, device=DEV)
        items.append((dy, x, dw, db, False))
        flops += 2 * M * n_in * n_out
    return items, flops


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('MVAE_HIP_LIB', 'default'))
    cases = [('image-dec', 1024, [(64, 512), (512, 512), (512, 512), (512, 784)]),
             ('label-dec', 1024, [(64, 512), (512, 512), (512, 512), (512, 10)]),
             ('img-enc', 512, [(784, 512), (512, 512), (512, 128)])]
    flush = torch.empty(128 << 20, dtype=torch.float32, device=DEV)
    for name, M, layers in cases:
        items, flops = batch(M, layers)
        for _ in range(3):
            K.linear_wgrad_batched(items)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    K.linear_wgrad_batched(items)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record(); g.replay(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20.0 * 1e3)
            cold = []
            for _ in range(8):
                flush.fill_(1.0)
                e0.record(); K.linear_wgrad_batched(items); e1.record()
                torch.cuda.synchronize()
                cold.append(e0.elapsed_time(e1) * 1e3)
        torch.cuda.current_stream().wait_stream(side)
        cold.sort()
        tf = flops / (best * 1e-6) / 1e12
        print('%-14s %-10s hot %7.2f us  cold(median) %7.2f us  %6.1f TFLOP/s  frac %.3f  (%.3f GFLOP)'
              % (tag, name, best, cold[len(cold) // 2], tf, tf / PEAK, flops / 1e9))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""One stack (encoder/decoder) of a model: forward_tape / backward_tape against torch autograd on the CPU,
layer by layer (diagnostic).  python tools/stack_probe.py celeba attrs_decoder 256 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mvae_amd  # noqa: E402
from mvae_amd import layers as L  # noqa: E402
from oracle import models as OM  # noqa: E402


def rel(a, b):
    b = b.double()
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    kind, stack, B, G = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    cls, d = OM.MODELS[kind]
    oracle = OM.fill_parameters(cls(d), 37).train()
    model = getattr(mvae_amd, kind).model.MVAE(d)
    model.load_state_dict(oracle.state_dict())
    model.cuda().train(); model.finalize()
    mod = getattr(model, stack)
    omod = getattr(oracle, stack)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(G * B, d, generator=g)
    plan = mod.plan()
    out, tape = L.forward_tape(plan, z.cuda(), groups=G)
    # oracle: run each group separately (BatchNorm statistics per group), collect per-layer outputs
    zs = z.clone().requires_grad_(True)
    outs = []
    inter = [[] for _ in range(G)]
    hooks = []
    seq = [m for m in omod.modules() if not list(m.children())]
    for gi in range(G):
        h = zs[gi * B:(gi + 1) * B]
        for m in omod.net:
            h = m(h)
            inter[gi].append(h)
        outs.append(h)
    ref = torch.cat(outs)
    print('output', rel(out, ref.detach()))
    # per-op saved inputs of the tape vs oracle intermediates
    li = -1
    names = [type(m).__name__ for m in omod.net]
    print(names)
    for i, (op, saved) in enumerate(zip(plan, tape)):
        print(i, op.kind, op.act, [None if s is None else tuple(s.shape) for s in saved])
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    gz = L.backward_tape(plan, tape, dy.cuda(), need_input_grad=True, groups=G)
    print('dz', rel(gz, zs.grad))
    og = dict(omod.named_parameters())
    for name, p in mod.named_parameters():
        r = og[name].grad
        print('%-20s %.3e (max|ref| %.2e)' % (name, rel(p.grad, r), r.abs().max().item()))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Per-parameter gradient error of one fused step against the CPU oracle (diagnostic).

    python tools/parity_probe.py celeba 256 [small_off]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MVAE_HIP_LIB', os.path.join(ROOT, 'multimodal-vae-public_amd', 'libmvae_hip_tuning.so'))

import torch  # noqa: E402

import mvae_amd  # noqa: E402
from mvae_amd import _lib  # noqa: E402
from mvae_amd.engine import BimodalStep  # noqa: E402
from oracle import models as OM, steps as OS  # noqa: E402


def main():
    kind, batch = sys.argv[1], int(sys.argv[2])
    if len(sys.argv) > 3 and sys.argv[3] == 'small_off':
        _lib.lib().mvae_debug_set_small(1, 0)
    cls, d = OM.MODELS[kind]
    oracle = OM.fill_parameters(cls(d), 37).train()
    model = getattr(mvae_amd, kind).model.MVAE(d)
    model.load_state_dict(oracle.state_dict())
    model.cuda().train(); model.finalize()
    image, label = OS.synthetic_batch(kind, batch, seed=91)
    torch.manual_seed(7)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))
    lam = 10.0 if kind == 'celeba' else 50.0
    total, terms, lat = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, lam, 0.5)
    total.backward()
    eng = BimodalStep(model, batch, 1.0, lam)
    elbo = eng.terms_in_reference_order(eng.step(image.cuda(), label.cuda(), 0.5, noise=noise)).cpu()
    print('terms', elbo.tolist(), [t.item() for t in terms])
    og = dict(oracle.named_parameters())
    for name, p in model.named_parameters():
        ref = og[name].grad
        err = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        print('%-44s %-22s max|ref| %.3e  rel err %.3e %s' % (name, tuple(ref.shape), ref.abs().max().item(), err,
                                                            '<<<' if err > 1e-4 else ''))


if __name__ == '__main__':
    main()

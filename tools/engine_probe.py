#!/usr/bin/env python
"""Intermediates of the fused CelebA step (label decoder branch) against torch on the CPU (diagnostic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mvae_amd  # noqa: E402
from mvae_amd.engine import BimodalStep  # noqa: E402
from oracle import models as OM, steps as OS  # noqa: E402


def rel(a, b):
    b = b.double()
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    kind, batch = 'celeba', int(sys.argv[1])
    cls, d = OM.MODELS[kind]
    oracle = OM.fill_parameters(cls(d), 37).train()
    model = getattr(mvae_amd, kind).model.MVAE(d)
    model.load_state_dict(oracle.state_dict())
    model.cuda().train(); model.finalize()
    image, label = OS.synthetic_batch(kind, batch, seed=91)
    torch.manual_seed(7)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=True)
    eng = BimodalStep(model, batch, 1.0, 10.0)
    eng.step(image.cuda(), label.cuda(), 0.5, noise=noise)
    torch.cuda.synchronize()
    z, kl, rows_lbl, g_lbl, rows_img, g_img, lbl_in, keep_dec = eng._carry['keep']
    _, tape_dl, dlog_lbl, _, tape_di, dlog_img = keep_dec
    logits_img, logits_lbl = eng.recon_logits()
    B = batch
    zc = z.cpu()
    coef = eng.coef.cpu()
    print('coef', coef)
    dec = oracle.attrs_decoder
    tot_ref_g = []
    for gi in range(3):
        zi = zc[gi].clone().requires_grad_(True)
        lo = dec(zi)
        print('group', gi, 'logits', rel(logits_lbl[gi * B:(gi + 1) * B], lo.detach()))
        dl = coef[1, gi] * (torch.sigmoid(lo.detach()) - label)
        print('   dlogits', rel(dlog_lbl[gi * B:(gi + 1) * B], dl) if dl.abs().max() > 0 else
              dlog_lbl[gi * B:(gi + 1) * B].abs().max().item())
        if dl.abs().max() > 0:
            diff = (dlog_lbl[gi * B:(gi + 1) * B].cpu() - dl).abs()
            bad = (diff > 1e-5 * dl.abs().max()).nonzero()
            print('   bad entries', bad.shape[0], 'first', bad[:6].tolist(), 'rows', sorted(set(bad[:, 0].tolist()))[:20])
            if bad.shape[0]:
                r, c = bad[0].tolist()
                print('   got', dlog_lbl[gi * B + r].cpu().tolist()[:6], 'ref', dl[r].tolist()[:6])
        lo.backward(dlog_lbl[gi * B:(gi + 1) * B].cpu())
        tot_ref_g.append(zi.grad)
    og = dict(dec.named_parameters())
    for name, p in model.attrs_decoder.named_parameters():
        print('%-16s %.3e' % (name, rel(p.grad, og[name].grad)))
    # tape inputs of each Linear vs oracle? (forward activations)
    for i, saved in enumerate(tape_dl):
        if saved is not None and saved[0] is not None:
            print('tape', i, tuple(saved[0].shape), float(saved[0].abs().max()))


if __name__ == '__main__':
    main()

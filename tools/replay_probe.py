#!/usr/bin/env python
"""GPU box: where does a replayed step differ from the oracle?  Replays one captured step, then runs an EAGER step
of a second engine on exactly the noise the graph drew, and the oracle on the same noise:
    python tools/replay_probe.py celeba 256 [weight_seed] [input_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

from mvae_amd.engine import BimodalStep  # noqa: E402
from mvae_amd.optim import FusedAdam  # noqa: E402
from oracle import steps as OS  # noqa: E402
from test_engine_gpu import build_pair  # noqa: E402

kind, batch = sys.argv[1], int(sys.argv[2])
wseed = int(sys.argv[3]) if len(sys.argv) > 3 else 53
iseed = int(sys.argv[4]) if len(sys.argv) > 4 else 191
lam = 10.0 if kind == 'celeba' else 50.0
DEV = 'cuda'

oracle, model, d = build_pair(kind, wseed)
opt = FusedAdam(model.parameters(), lr=1e-3)
eng = BimodalStep(model, batch, 1.0, lam, seed=77)
image, label = OS.synthetic_batch(kind, batch, seed=iseed)
eng.capture(opt, image.shape[1:], label)
eng.replay(image.to(DEV), label.to(DEV), 0.5)
torch.cuda.synchronize()
g_replay = model.arena.grad.clone()
noise = {'eps': [eng.noise[eng.ref_order.index(r)].cpu() for r in range(3)], 'mask': [None] * 3}
if eng.drop_masks is not None:
    noise['mask'][0], noise['mask'][1] = eng.drop_masks[0].cpu(), eng.drop_masks[1].cpu()
logits_img, logits_lbl = eng.recon_logits()
li_replay = logits_img.clone()

_, model2, _ = build_pair(kind, wseed)
eng2 = BimodalStep(model2, batch, 1.0, lam, seed=77)
eng2.step(image.to(DEV), label.to(DEV), 0.5, noise=noise)
torch.cuda.synchronize()
g_eager = model2.arena.grad.clone()
li_eager = eng2.recon_logits()[0]

total, terms, lat = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, lam, 0.5)
total.backward()
print('replay vs eager: max |dg| / max|g| = %.3e ; logits max abs diff %.3e' % (
    (g_replay - g_eager).abs().max().item() / g_eager.abs().max().item(), (li_replay - li_eager).abs().max().item()))
flat0 = model2.arena.flat.data_ptr()
og = dict(oracle.named_parameters())
for name, p in model2.named_parameters():
    off = (p.data_ptr() - flat0) // 4
    ref = og[name].grad
    sc = ref.abs().max().item() + 1e-30
    e1 = (g_eager[off:off + p.numel()].view(p.shape).cpu() - ref).abs().max().item() / sc
    e2 = (g_replay[off:off + p.numel()].view(p.shape).cpu() - ref).abs().max().item() / sc
    if max(e1, e2) > 5e-5:
        print('%-40s eager-vs-oracle %.2e  replay-vs-oracle %.2e' % (name, e1, e2))
# the oracle's own image logits of call 1: exact zeros / sign flips against the HIP ones
with torch.no_grad():
    ri, rl, mu, lv, z = oracle(image, label, eps=noise['eps'][0], dropout_mask=noise['mask'][0])
hip = li_eager.reshape(-1, ri[0].numel())[:batch].cpu().reshape(ri.shape)
print('oracle exact-zero logits (call 1): %d ; HIP exact zeros: %d ; sign flips: %d ; max |diff| %.3e' % (
    int((ri == 0).sum()), int((hip == 0).sum()), int(((ri > 0) != (hip > 0)).sum()), (ri - hip).abs().max().item()))
small = (ri.abs() < 1e-6)
print('oracle logits with |x| < 1e-6: %d' % int(small.sum()))

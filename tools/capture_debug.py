#!/usr/bin/env python
"""GPU box: where does a captured reference body differ from the eager one?  per-parameter max |diff| after 1 and 2 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import mvae_amd
from test_capture_step_gpu import _run
from mvae_amd.optim import FusedAdam

kind, batch = sys.argv[1], int(sys.argv[2])
mk = lambda ps: FusedAdam(ps, lr=1e-3)
for n in (1, 2):
    l_e, st_e, w_e, _ = _run(kind, batch, False, mk, n_steps=n)
    l_c, st_c, w_c, _ = _run(kind, batch, True, mk, n_steps=n)
    print('steps', n, 'losses', l_e, l_c)
    for k in st_e:
        d = (st_e[k].float() - st_c[k].float()).abs().max().item()
        if d > 0:
            print('  %-50s max|diff| %.3e  max|value| %.3e' % (k, d, st_e[k].float().abs().max().item()))
# eager vs eager: is the eager path itself reproducible?
l_a, st_a, _, _ = _run(kind, batch, False, mk, n_steps=2)
l_b, st_b, _, _ = _run(kind, batch, False, mk, n_steps=2)
print('eager vs eager:', l_a == l_b, max((st_a[k].float() - st_b[k].float()).abs().max().item() for k in st_a))

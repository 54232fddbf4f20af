"""Shared helpers for the parity tests."""
import ast
import os

import numpy as np
import torch

REL_TOL = 1e-4   # north_star: outputs (ELBO, gradients) within 1e-4 relative, fp32


def load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    meta = ast.literal_eval(str(fx.pop('meta')))
    return fx, meta


def rel_err(a, b):
    """max |a-b| / max |b| -- the elementwise criterion of SURVEY Appendix D (1e-4 * max|g|)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().double().cpu()
    assert a.shape == b.shape, 'shape %s vs %s' % (tuple(a.shape), tuple(b.shape))
    if b.numel() == 0:
        return 0.0
    den = b.abs().max().item()
    return (a - b).abs().max().item() / max(den, 1e-30)


def assert_close(a, b, what, tol=REL_TOL):
    e = rel_err(a, b)
    assert e <= tol, '%s: relative error %.3e > %.1e' % (what, e, tol)
    return e


def golden_noise(fx, n_calls):
    noise = {'eps': [], 'mask': []}
    for c in range(n_calls):
        noise['eps'].append(torch.from_numpy(fx['eps%d' % c]))
        k = 'mask%d' % c
        noise['mask'].append(torch.from_numpy(fx[k]).float() if k in fx else None)
    return noise


# Gradients that are mathematically ZERO: the bias of a Linear that feeds a training-mode BatchNorm
# (celeba/model.py:148-151,175-181 -- the batch mean is subtracted right behind it).  The reference returns
# pure round-off there (~1e-7 of the layer's weight gradient), so these tensors -- and only these -- are
# compared on an absolute floor of GRAD_FLOOR x the model's largest gradient; every other parameter is
# compared on its own magnitude.
ZERO_GRAD_PARAMS = {
    'celeba': ('attrs_encoder.net.0.bias', 'attrs_encoder.net.3.bias',
               'attrs_decoder.net.0.bias', 'attrs_decoder.net.3.bias', 'attrs_decoder.net.6.bias'),
}
GRAD_FLOOR = 1e-2


def grad_floor(kind, name, global_scale):
    """Absolute comparison floor for parameter ``name`` of model ``kind`` (0 for all but the named tensors)."""
    return GRAD_FLOOR * global_scale if name in ZERO_GRAD_PARAMS.get(kind, ()) else 0.0

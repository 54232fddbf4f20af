"""autograd bridges for the reference's *function* surface (``elbo_loss`` and friends) and for
the latent path of ``MVAE.forward``.  Every ``Function`` is a thin shell over HIP launches in
``kernels.py``; shapes, argument meaning and error behaviour follow the reference:

    binary_cross_entropy_with_logits  mnist/train.py:62-74   (ValueError on size mismatch)
    cross_entropy                     mnist/train.py:77-94
    elbo_loss (bimodal, label)        mnist/train.py:20-59, fashionmnist/train.py:20-59
    elbo_loss (bimodal, attributes)   celeba/train.py:22-65
    elbo_loss (N-modal lists)         celeba19/train.py:26-60

The fused train step (``engine.py``) does not go through these: it launches the same kernels
directly with the loss gradient folded into the forward pass.
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib
from . import kernels as K


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------- latent path
class PoEFn(torch.autograd.Function):
    """(heads_0 .. heads_{E-1}) -> (mu, logvar, z, kl) for T terms.  ``heads_e`` is an encoder
    output [B, 2D] (mu | logvar) or a (mu, logvar) pair given as two [B, D] tensors sharing a
    row stride."""

    @staticmethod
    def forward(ctx, cfg, *heads):
        masks_dev, noise, variant, D = cfg
        mus = [h[:, :D] for h in heads]
        lvs = [h[:, D:] for h in heads]
        T = masks_dev.numel()
        B = heads[0].shape[0]
        dev = heads[0].device
        mu = torch.empty(T, B, D, dtype=torch.float32, device=dev)
        lv = torch.empty_like(mu)
        z = torch.empty_like(mu)
        kl = torch.empty(T, B, dtype=torch.float32, device=dev)
        K.poe_fwd(mus, lvs, masks_dev, noise, mu, lv, z, kl, variant)
        ctx.cfg = cfg
        ctx.save_for_backward(mu, lv, *heads)
        return mu, lv, z, kl

    @staticmethod
    def backward(ctx, dmu, dlv, dz, dkl):
        masks_dev, noise, variant, D = ctx.cfg
        mu, lv = ctx.saved_tensors[:2]
        heads = ctx.saved_tensors[2:]
        mus = [h[:, :D] for h in heads]
        lvs = [h[:, D:] for h in heads]
        grads = [torch.empty_like(h) for h in heads]
        g_mus = [g[:, :D] for g in grads]
        g_lvs = [g[:, D:] for g in grads]

        def c(t):
            return None if t is None else t.contiguous()
        K.poe_bwd(mus, lvs, masks_dev, noise, mu, lv, c(dz), c(dmu), c(dlv), c(dkl), g_mus, g_lvs, variant)
        return (None,) + tuple(grads)


class PoEStackFn(torch.autograd.Function):
    """``ProductOfExperts.forward`` of the reference on a stacked [M, B, D] pair (mnist/model.py:156-163,
    celeba/model.py:200-207): the product over ALL M rows (no built-in prior -- row 0 of the reference's stack is
    what ``prior_expert`` returned), one launch forward, one backward."""

    @staticmethod
    def forward(ctx, mu_stack, lv_stack, variant):
        mu_stack, lv_stack = mu_stack.contiguous(), lv_stack.contiguous()
        M, B, D = mu_stack.shape
        dev = mu_stack.device
        masks = K.all_experts_mask(M, dev)
        mu = torch.empty(1, B, D, dtype=torch.float32, device=dev)
        lv = torch.empty_like(mu)
        mus = [mu_stack[e] for e in range(M)]
        lvs = [lv_stack[e] for e in range(M)]
        K.poe_fwd(mus, lvs, masks, None, mu, lv, None, None, variant + '-noprior')
        ctx.variant = variant
        ctx.save_for_backward(mu, lv, mu_stack, lv_stack, masks)
        return mu[0], lv[0]

    @staticmethod
    def backward(ctx, dmu, dlv):
        mu, lv, mu_stack, lv_stack, masks = ctx.saved_tensors
        M = mu_stack.shape[0]
        g_mu, g_lv = torch.empty_like(mu_stack), torch.empty_like(lv_stack)
        K.poe_bwd([mu_stack[e] for e in range(M)], [lv_stack[e] for e in range(M)], masks, None, mu, lv, None,
                  dmu.contiguous().reshape(mu.shape), dlv.contiguous().reshape(lv.shape), None,
                  [g_mu[e] for e in range(M)], [g_lv[e] for e in range(M)], ctx.variant + '-noprior')
        return g_mu, g_lv, None


class ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        z = torch.empty_like(mu)
        _lib.check(_lib.lib().mvae_reparam_fwd(K._ptr(mu), K._ptr(logvar), K._ptr(eps), K._ptr(z), mu.numel(),
                                               _stream()), 'mvae_reparam_fwd')
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = dz.contiguous()
        dmu, dlv = torch.empty_like(dz), torch.empty_like(dz)
        _lib.check(_lib.lib().mvae_reparam_bwd(K._ptr(dz), K._ptr(logvar), K._ptr(eps), K._ptr(dmu), K._ptr(dlv),
                                               dz.numel(), _stream()), 'mvae_reparam_bwd')
        return dmu, dlv, None


# ----------------------------------------------------------------------------- loss pieces
class _BceElemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        xc, tc = x.contiguous(), t.contiguous()
        out = torch.empty_like(xc)
        K.bce_elem_fwd(xc, tc, out)
        ctx.save_for_backward(xc, tc)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.bce_elem_bwd(x, t, g.contiguous(), dx)
        return dx.view(g.shape), None


class _BceRowsFn(torch.autograd.Function):
    """rows[r] = sum_p bce(x[r,p], t[r,p]) -- image BCE summed over pixels in one pass."""

    @staticmethod
    def forward(ctx, x, t):
        rows = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        K.bce_rowsum_fwd(x, t, rows, rows_per_group=1)
        ctx.save_for_backward(x, t)
        return rows

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.bce_rowsum_bwd(x, t, g.contiguous(), dx, rows_per_group=1)
        return dx, None


class _CeRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        rows = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        K.ce_fwd(x, y, rows)
        ctx.save_for_backward(x, y)
        return rows

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.ce_bwd(x, y, g.contiguous(), dx, rows_per_group=1)
        return dx, None


class _KlRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        kl = torch.empty(mu.shape[0], dtype=torch.float32, device=mu.device)
        K.kl_rows_fwd(mu, logvar, kl)
        ctx.save_for_backward(mu, logvar)
        return kl

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        K.kl_rows_bwd(mu, logvar, g.contiguous(), dmu, dlv)
        return dmu, dlv


# bounded: an annealing schedule hands elbo_loss a new factor every step (mnist/train.py:184-194: ~10^5 distinct values
# over a run), and each entry pins a device allocation
_COEF_CACHE = collections.OrderedDict()
_COEF_CACHE_MAX = 128


def _w(x):
    """A loss weight as the reference passes it (a python number), or -- inside ``mvae_amd.capture_step`` -- the 0-d
    device tensor the captured graph re-reads at every replay (the annealing factor changes from step to step)."""
    return x if torch.is_tensor(x) else float(x)


def _coef_tensor(weights, B, dev):
    """[n] fp32 device tensor of w_i / B.  Python numbers come from a cache per (values, B, device): no host-to-device
    copy per ``elbo_loss`` call, and nothing a stream capture would refuse; a 0-d device tensor among the weights is spliced
    in on the device."""
    consts = tuple(None if torch.is_tensor(w) else float(w) for w in weights)
    key = (consts, int(B), str(dev))
    base = _COEF_CACHE.get(key)
    if base is not None:
        _COEF_CACHE.move_to_end(key)
    else:
        # w * (1 / B), both in fp32 -- the arithmetic the device applies to a tensor weight below (a MULTIPLY by the rounded
        # reciprocal: torch's division of a CUDA tensor by a host scalar is one too) -- so that a python number and the same
        # number arriving as a device scalar (capture_step) give the same bits (0.6 / 6 rounds differently in each of:
        # double division, fp32 division, fp32 multiply by 1/6)
        inv_b = np.float32(1.0) / np.float32(B)
        vals = [float(np.float32(c or 0.0) * inv_b) for c in consts]
        if torch.device(dev).type == 'cuda' and torch.cuda.is_current_stream_capturing():     # (a pageable-memory upload is not capturable: fills are)
            base = torch.stack([torch.full((), v, dtype=torch.float32, device=dev) for v in vals])
        else:
            base = torch.tensor(vals, dtype=torch.float32, device=dev)
            _COEF_CACHE[key] = base
            if len(_COEF_CACHE) > _COEF_CACHE_MAX:
                _COEF_CACHE.popitem(last=False)       # least recently used; a graph or an autograd node keeps its own reference
    if all(c is not None for c in consts):
        return base
    out = base.clone()
    for i, w in enumerate(weights):
        if torch.is_tensor(w):
            out[i:i + 1].copy_(w.detach().reshape(1).to(torch.float32) * float(np.float32(1.0) / np.float32(B)))
    return out


class _WeightedMeanFn(torch.autograd.Function):
    """mean_b(sum_i w_i * rows_i[b]) for up to three row vectors (the last line of elbo_loss)."""

    @staticmethod
    def forward(ctx, weights, *rows):
        B = rows[0].shape[0]
        dev = rows[0].device
        out = torch.empty(1, dtype=torch.float32, device=dev)
        coefs = _coef_tensor(weights, B, dev)
        for i, r in enumerate(rows):
            K.group_sums(r.contiguous(), coefs[i:i + 1], None, out, 1, B, accumulate=(i > 0))
        ctx.coefs, ctx.B = coefs, B
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple((g * ctx.coefs[i]).expand(ctx.B) for i in range(ctx.coefs.numel()))


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('multimodal-vae-public_amd losses run on the GPU only (tensor on %s)' % t.device)


# ----------------------------------------------------------------------------- reference surface
def binary_cross_entropy_with_logits(input, target):
    """Elementwise sigmoid + BCE (mnist/train.py:62-74)."""
    if not (target.size() == input.size()):
        raise ValueError("Target size ({}) must be the same as input size ({})".format(
            target.size(), input.size()))
    _need_gpu(input)
    return _BceElemFn.apply(input, target)


def cross_entropy(input, target, eps=1e-6):
    """[N, K] matrix holding -log_softmax(input + 1e-6) at the label column, 0 elsewhere
    (mnist/train.py:77-94); ``eps`` is fixed at the reference's default."""
    if not (target.size(0) == input.size(0)):
        raise ValueError(
            "Target size ({}) must be the same as input size ({})".format(
                target.size(0), input.size(0)))
    if eps != 1e-6:
        raise ValueError('cross_entropy is built for the reference default eps=1e-6')
    _need_gpu(input)
    rows = _CeRowsFn.apply(input.contiguous(), target.contiguous())
    out = torch.zeros_like(input)
    return out.scatter(1, target.unsqueeze(1), rows.unsqueeze(1))


def _image_rows(recon, image):
    n = image[0].numel()
    x = recon.reshape(-1, n).contiguous()
    t = image.reshape(-1, n).contiguous()
    if x.shape != t.shape:
        raise ValueError("Target size ({}) must be the same as input size ({})".format(
            t.size(), x.size()))
    return _BceRowsFn.apply(x, t)


def elbo_loss_label(recon_image, image, recon_text, text, mu, logvar,
                    lambda_image=1.0, lambda_text=1.0, annealing_factor=1):
    """mnist/train.py:20-59 (same text in fashionmnist/train.py)."""
    _need_gpu(mu)
    rows, weights = [], []
    if recon_image is not None and image is not None:
        rows.append(_image_rows(recon_image, image)); weights.append(float(lambda_image))
    if recon_text is not None and text is not None:
        if not (text.size(0) == recon_text.size(0)):
            raise ValueError("Target size ({}) must be the same as input size ({})".format(
                text.size(0), recon_text.size(0)))
        rows.append(_CeRowsFn.apply(recon_text.contiguous(), text.contiguous()))
        weights.append(float(lambda_text))
    rows.append(_KlRowsFn.apply(mu, logvar)); weights.append(_w(annealing_factor))
    return _WeightedMeanFn.apply(weights, *rows)


def elbo_loss_attrs(recon_image, image, recon_attrs, attrs, mu, logvar,
                    lambda_image=1.0, lambda_attrs=1.0, annealing_factor=1):
    """celeba/train.py:22-65: the 18 per-column BCEs are one row-sum launch."""
    _need_gpu(mu)
    rows, weights = [], []
    if recon_image is not None and image is not None:
        rows.append(_image_rows(recon_image, image)); weights.append(float(lambda_image))
    if recon_attrs is not None and attrs is not None:
        if not (attrs.size() == recon_attrs.size()):
            raise ValueError("Target size ({}) must be the same as input size ({})".format(
                attrs[:, 0].size(), recon_attrs[:, 0].size()))
        rows.append(_BceRowsFn.apply(recon_attrs.contiguous(), attrs.contiguous()))
        weights.append(float(lambda_attrs))
    rows.append(_KlRowsFn.apply(mu, logvar)); weights.append(_w(annealing_factor))
    return _WeightedMeanFn.apply(weights, *rows)


def elbo_loss_multi(recon, data, mu, logvar, lambda_image=1.0, lambda_attrs=1.0,
                    annealing_factor=1.):
    """celeba19/train.py:26-60: list entries with more than one dim are images (:52-55),
    1-D entries are single attributes (:56-57)."""
    assert len(recon) == len(data), "must supply ground truth for every modality."
    _need_gpu(mu)
    batch_size = mu.size(0)
    rows, weights = [], []
    attr_x, attr_t = [], []
    for ix in range(len(recon)):
        if recon[ix].dim() > 1:
            rows.append(_image_rows(recon[ix].reshape(batch_size, -1), data[ix].reshape(batch_size, -1)))
            weights.append(float(lambda_image))
        else:
            if not (data[ix].size() == recon[ix].size()):
                raise ValueError("Target size ({}) must be the same as input size ({})".format(
                    data[ix].size(), recon[ix].size()))
            attr_x.append(recon[ix]); attr_t.append(data[ix])
    if attr_x:
        x = torch.stack(attr_x, dim=1).contiguous()
        t = torch.stack(attr_t, dim=1).contiguous().float()
        rows.append(_BceRowsFn.apply(x, t)); weights.append(float(lambda_attrs))
    rows.append(_KlRowsFn.apply(mu, logvar)); weights.append(_w(annealing_factor))
    return _WeightedMeanFn.apply(weights, *rows)

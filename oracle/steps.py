"""Oracle (test infrastructure): the body of the reference's ``train(epoch)`` closures.

``bimodal_step``  -- mnist/train.py:197-218, fashionmnist/train.py:197-218,
                     celeba/train.py:190-212 (three forwards, three ELBOs, sum).
``celeba19_step`` -- celeba19/train.py:257-308 (complete + image + 18 single-attribute
                     + M sampled-subset ELBOs).
Noise is drawn (``draw_*_noise``) in the order the reference's global generator
would produce it, so a run of the real reference under the same
``torch.manual_seed`` sees identical values.
"""
import numpy as np
import torch

from . import functional as OF
from .models import N_ATTRS


# ----------------------------------------------------------------------------
# noise, in reference draw order
# ----------------------------------------------------------------------------
def draw_bimodal_noise(batch, n_latents, has_dropout, generator=None):
    """Per step: call 1 (image+label), call 2 (image), call 3 (label).  Each call
    draws [dropout mask [B,512] if the image encoder runs and has Dropout] then
    eps [B,D] (celeba/model.py:91 then :29-33)."""
    noise = {'eps': [], 'mask': []}
    for has_image in (True, True, False):
        if has_image and has_dropout:
            noise['mask'].append(torch.empty(batch, 512).bernoulli_(0.9, generator=generator))
        else:
            noise['mask'].append(None)
        noise['eps'].append(torch.empty(batch, n_latents).normal_(generator=generator))
    return noise


def celeba19_terms(sample_combos):
    """The ELBO terms of one celeba19 step as ``(present[19] bool, use_lambdas)``:
    complete, image-only, 18 single attributes, then the sampled subsets
    (celeba19/train.py:264-302).  ``use_lambdas`` is False where the reference
    omits ``lambda_image/lambda_attrs`` and so uses 1.0 (:281-282, :294-300)."""
    terms = [(np.ones(1 + N_ATTRS, dtype=bool), True)]
    img_only = np.zeros(1 + N_ATTRS, dtype=bool); img_only[0] = True
    terms.append((img_only, True))
    for ix in range(N_ATTRS):
        m = np.zeros(1 + N_ATTRS, dtype=bool); m[1 + ix] = True
        terms.append((m, False))
    for combo in sample_combos:
        terms.append((np.asarray(combo, dtype=bool), False))
    return terms


def draw_celeba19_noise(batch, n_latents, terms, generator=None):
    noise = {'eps': [], 'mask': []}
    for present, _ in terms:
        if present[0]:
            noise['mask'].append(torch.empty(batch, 512).bernoulli_(0.9, generator=generator))
        else:
            noise['mask'].append(None)
        noise['eps'].append(torch.empty(batch, n_latents).normal_(generator=generator))
    return noise


# ----------------------------------------------------------------------------
# celeba19 subset sampling (host side) -- celeba19/train.py:87-142
# ----------------------------------------------------------------------------
def enumerate_combinations(n):
    """All subsets of size 2..n-1 of n modalities as a bool matrix
    (celeba19/train.py:87-108).  Built with bit tricks instead of itertools;
    row order matches the reference (by size, then lexicographic)."""
    from itertools import combinations
    rows = []
    for size in range(2, n):
        for combo in combinations(range(n), size):
            r = np.zeros(n, dtype=bool)
            r[list(combo)] = True
            rows.append(r)
    return np.stack(rows)


def sample_combinations(pool, size=1, rng=np.random):
    """Draw ``size`` subset sizes uniformly from the sizes present in the pool,
    then that many distinct subsets of each size (celeba19/train.py:111-142)."""
    n_modalities = pool.shape[1]
    pool_sums = pool.sum(axis=1)
    pool_dist = np.bincount(pool_sums)
    pool_space = np.where(pool_dist > 0)[0]
    sample_pool = rng.choice(pool_space, size, replace=True)
    sample_dist = np.bincount(sample_pool)
    if sample_dist.size < n_modalities:
        sample_dist = np.concatenate(
            (sample_dist, np.zeros(n_modalities - sample_dist.size, dtype=int)))
    out = []
    for ix in range(n_modalities):
        if sample_dist[ix] > 0:
            pool_i = pool[pool_sums == ix]
            pick = rng.choice(range(pool_i.shape[0]), size=sample_dist[ix], replace=False)
            out.append(pool_i[pick])
    return np.concatenate(out)


# ----------------------------------------------------------------------------
# steps
# ----------------------------------------------------------------------------
def bimodal_step(model, kind, image, label, noise, lambda_image, lambda_label,
                 annealing_factor, return_recon=False):
    """One train-step loss (no optimizer): returns (total, [joint, image, label] terms,
    [(mu, logvar, z)] per call); with ``return_recon`` also the reconstructions that enter a
    loss, [(image logits, label logits)] per call (None where the reference passes None,
    mnist/train.py:208,211)."""
    elbo = OF.elbo_loss_attrs if kind == 'celeba' else OF.elbo_loss_label
    ri1, rl1, mu1, lv1, z1 = model(image, label, eps=noise['eps'][0], dropout_mask=noise['mask'][0])
    ri2, rl2, mu2, lv2, z2 = model(image, None, eps=noise['eps'][1], dropout_mask=noise['mask'][1])
    ri3, rl3, mu3, lv3, z3 = model(None, label, eps=noise['eps'][2], dropout_mask=noise['mask'][2])
    joint = elbo(ri1, image, rl1, label, mu1, lv1, lambda_image, lambda_label, annealing_factor)
    img = elbo(ri2, image, None, None, mu2, lv2, lambda_image, lambda_label, annealing_factor)
    lbl = elbo(None, None, rl3, label, mu3, lv3, lambda_image, lambda_label, annealing_factor)
    total = joint + img + lbl
    latents = [(mu1, lv1, z1), (mu2, lv2, z2), (mu3, lv3, z3)]
    if return_recon:
        return total, [joint, img, lbl], latents, [(ri1, rl1), (ri2, None), (None, rl3)]
    return total, [joint, img, lbl], latents


def celeba19_step(model, image, attrs2d, terms, noise, lambda_image, lambda_attrs,
                  annealing_factor):
    """``attrs2d`` is ``[B, 18]`` float; split into a list like tensor_2d_to_list
    (celeba19/train.py:78-84)."""
    attrs = [attrs2d[:, i] for i in range(N_ATTRS)]
    total = 0
    elbos, latents = [], []
    for t, (present, use_lambdas) in enumerate(terms):
        img_in = image if present[0] else None
        attrs_in = [attrs[i] if present[1 + i] else None for i in range(N_ATTRS)]
        ri, ra, mu, lv, z = model(img_in, attrs_in, eps=noise['eps'][t],
                                  dropout_mask=noise['mask'][t])
        recon, data = [], []
        if present[0]:
            recon.append(ri); data.append(image)
        for i in range(N_ATTRS):
            if present[1 + i]:
                recon.append(ra[i]); data.append(attrs[i])
        if use_lambdas:
            e = OF.elbo_loss_multi(recon, data, mu, lv, lambda_image=lambda_image,
                                   lambda_attrs=lambda_attrs,
                                   annealing_factor=annealing_factor)
        else:
            e = OF.elbo_loss_multi(recon, data, mu, lv, annealing_factor=annealing_factor)
        total = total + e
        elbos.append(e)
        latents.append((mu, lv, z))
    return total, elbos, latents


def synthetic_batch(kind, batch, seed):
    """Random-pixel / random-label batch of SURVEY.md section 8(d)."""
    g = torch.Generator().manual_seed(seed)
    if kind in ('mnist', 'fashionmnist'):
        image = torch.rand(batch, 1, 28, 28, generator=g)
        label = torch.randint(0, 10, (batch,), generator=g)
    else:
        image = torch.rand(batch, 3, 64, 64, generator=g)
        label = torch.randint(0, 2, (batch, N_ATTRS), generator=g).float()
    return image, label

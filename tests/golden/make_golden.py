#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

For each hot-path experiment it imports the reference's ``model.py`` and
``train.py`` (import-time shims only: a stub torchvision / datasets module and
the py2 names ``xrange``, ``np.bool``, ``np.int`` -- SURVEY.md Appendix C),
loads the deterministic weights of ``oracle.models.fill_parameters``, runs ONE
train-step body exactly as the reference's ``train(epoch)`` closure does
(three / 20+M ``model()`` calls, the reference's own ``elbo_loss``, ``backward``)
under ``torch.manual_seed(noise_seed)``, and records:

  inputs, the noise the global generator produced (replayed in draw order),
  per-term ELBOs, total loss, mu/logvar/z per call, image/label logits of
  sample 0 of call 1, per-parameter gradient L2 norm + first 8 elements, and
  the BatchNorm running statistics after the step.

It also asserts that the oracle restatement reproduces every recorded value,
so a fixture is only ever written from a state where oracle == reference.
The fixtures are data (npz); no reference source text is stored.
"""
import builtins
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import models as OM, steps as OS  # noqa: E402

WEIGHT_SEED = 7
LAMBDA_IMAGE = 1.0
LAMBDA_LABEL = {'mnist': 50.0, 'fashionmnist': 50.0, 'celeba': 10.0, 'celeba19': 10.0}
BETA = 0.5


def import_reference(exp):
    builtins.xrange = range
    np.bool = bool
    np.int = int
    tv = types.ModuleType('torchvision')
    tv.transforms = types.ModuleType('torchvision.transforms')
    tv.datasets = types.ModuleType('torchvision.datasets')
    tv.datasets.MNIST = object
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tv.transforms,
                        'torchvision.datasets': tv.datasets})
    for name in ('model', 'train', 'datasets'):
        sys.modules.pop(name, None)
    if exp in ('celeba', 'celeba19'):
        ds = types.ModuleType('datasets')
        ds.N_ATTRS = 18
        ds.CelebAttributes = object
        sys.modules['datasets'] = ds
    path = os.path.join(REF, exp)
    sys.path.insert(0, path)
    try:
        model = importlib.import_module('model')
        train = importlib.import_module('train')
    finally:
        sys.path.remove(path)
    return model, train


def grad_digest(model):
    out = {}
    for name, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        out['gnorm/' + name] = np.float64(g.double().norm().item())
        out['ghead/' + name] = g[:8].numpy().copy()
    return out


def bn_stats(model):
    return {'bn/' + k: v.detach().numpy().copy() for k, v in model.state_dict().items()
            if 'running_' in k or 'num_batches' in k}


def check(a, b, what, rtol=2e-5, atol=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    tol = atol + rtol * np.abs(b).max() if b.size else atol
    assert err <= tol, '%s: oracle differs from reference by %g (tol %g)' % (what, err, tol)


def run_bimodal(exp, batch, noise_seed):
    ref_model_mod, ref_train = import_reference(exp)
    cls, d = OM.MODELS[exp]
    oracle = OM.fill_parameters(cls(d), WEIGHT_SEED)
    ref = ref_model_mod.MVAE(d)
    ref.load_state_dict(oracle.state_dict())
    ref.train(); oracle.train()
    image, label = OS.synthetic_batch(exp, batch, seed=1234)
    lam_i, lam_l = LAMBDA_IMAGE, LAMBDA_LABEL[exp]

    # ---- the reference step body (mnist/train.py:197-218) ----
    torch.manual_seed(noise_seed)
    kw = 'attrs' if exp == 'celeba' else 'text'
    lkw = 'lambda_attrs' if exp == 'celeba' else 'lambda_text'
    r1 = ref(image, label)
    r2 = ref(image)
    r3 = ref(**{kw: label})
    joint = ref_train.elbo_loss(r1[0], image, r1[1], label, r1[2], r1[3],
                                lambda_image=lam_i, annealing_factor=BETA, **{lkw: lam_l})
    img = ref_train.elbo_loss(r2[0], image, None, None, r2[2], r2[3],
                              lambda_image=lam_i, annealing_factor=BETA, **{lkw: lam_l})
    lbl = ref_train.elbo_loss(None, None, r3[1], label, r3[2], r3[3],
                              lambda_image=lam_i, annealing_factor=BETA, **{lkw: lam_l})
    total = joint + img + lbl
    total.backward()

    # ---- replay the generator to recover the noise, run the oracle ----
    torch.manual_seed(noise_seed)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=(exp == 'celeba'))
    o_total, o_terms, o_lat = OS.bimodal_step(oracle, exp, image, label, noise,
                                              lam_i, lam_l, BETA)
    o_total.backward()

    fx = {'image': image.numpy(), 'label': label.numpy(),
          'total': np.float64(total.item()),
          'terms': np.array([joint.item(), img.item(), lbl.item()], dtype=np.float64),
          'logits_image_0': r1[0][0].detach().numpy().reshape(-1)[:64].copy(),
          'logits_label_0': r1[1][0].detach().numpy().copy()}
    for c, r in enumerate((r1, r2, r3)):
        fx['mu%d' % c] = r[2].detach().numpy().copy()
        fx['logvar%d' % c] = r[3].detach().numpy().copy()
        fx['eps%d' % c] = noise['eps'][c].numpy()
        if noise['mask'][c] is not None:
            fx['mask%d' % c] = noise['mask'][c].numpy().astype(np.uint8)
        check(o_lat[c][0].detach(), r[2].detach(), exp + ' mu%d' % c)
        check(o_lat[c][1].detach(), r[3].detach(), exp + ' logvar%d' % c)
        fx['z%d' % c] = o_lat[c][2].detach().numpy().copy()
    fx.update(grad_digest(ref))
    fx.update(bn_stats(ref))
    check(o_total.item(), total.item(), exp + ' total')
    check([t.item() for t in o_terms], fx['terms'], exp + ' terms')
    og = grad_digest(oracle)
    for k in og:
        check(og[k], fx[k], exp + ' ' + k, rtol=1e-4, atol=1e-7)
    ob = bn_stats(oracle)
    for k in ob:
        check(ob[k], fx[k], exp + ' ' + k)
    meta = dict(exp=exp, batch=batch, n_latents=d, weight_seed=WEIGHT_SEED,
                noise_seed=noise_seed, input_seed=1234, lambda_image=lam_i,
                lambda_label=lam_l, beta=BETA)
    return fx, meta


def run_celeba19(batch, noise_seed, approx_m=1):
    exp = 'celeba19'
    ref_model_mod, ref_train = import_reference(exp)
    cls, d = OM.MODELS[exp]
    oracle = OM.fill_parameters(cls(d), WEIGHT_SEED)
    ref = ref_model_mod.MVAE(d)
    ref.load_state_dict(oracle.state_dict())
    ref.train(); oracle.train()
    image, attrs2d = OS.synthetic_batch(exp, batch, seed=1234)
    lam_i, lam_a = LAMBDA_IMAGE, LAMBDA_LABEL[exp]

    # combination pool and sampling: the reference's own functions
    pool = ref_train.enumerate_combinations(19)
    o_pool = OS.enumerate_combinations(19)
    assert pool.shape == o_pool.shape == (524267, 19) and (pool == o_pool).all()
    np.random.seed(4321)
    combos = ref_train.sample_combinations(pool, size=approx_m)
    np.random.seed(4321)
    o_combos = OS.sample_combinations(o_pool, size=approx_m)
    assert (combos == o_combos).all()

    # ---- the reference step body (celeba19/train.py:254-308) ----
    torch.manual_seed(noise_seed)
    attrs = ref_train.tensor_2d_to_list(attrs2d)
    train_loss = 0
    elbos, lat = [], []
    recon_image, recon_attrs, mu, logvar = ref(image, attrs)
    e = ref_train.elbo_loss([recon_image] + recon_attrs, [image] + attrs, mu, logvar,
                            lambda_image=lam_i, lambda_attrs=lam_a, annealing_factor=BETA)
    train_loss += e; elbos.append(e); lat.append((mu, logvar))
    logits0 = recon_image[0].detach().numpy().reshape(-1)[:64].copy()
    recon_image, _, mu, logvar = ref(image=image)
    e = ref_train.elbo_loss([recon_image], [image], mu, logvar,
                            lambda_image=lam_i, lambda_attrs=lam_a, annealing_factor=BETA)
    train_loss += e; elbos.append(e); lat.append((mu, logvar))
    for ix in range(len(attrs)):
        _, recon_attrs, mu, logvar = ref(attrs=[attrs[k] if k == ix else None
                                                for k in range(len(attrs))])
        e = ref_train.elbo_loss([recon_attrs[ix]], [attrs[ix]], mu, logvar,
                                annealing_factor=BETA)
        train_loss += e; elbos.append(e); lat.append((mu, logvar))
    for sample_combo in combos:
        attrs_combo = sample_combo[1:]
        recon_image, recon_attrs, mu, logvar = ref(
            image=image if sample_combo[0] else None,
            attrs=[attrs[ix] if attrs_combo[ix] else None for ix in range(attrs_combo.size)])
        if sample_combo[0]:
            e = ref_train.elbo_loss(
                [recon_image] + [recon_attrs[ix] for ix in range(attrs_combo.size) if attrs_combo[ix]],
                [image] + [attrs[ix] for ix in range(attrs_combo.size) if attrs_combo[ix]],
                mu, logvar, annealing_factor=BETA)
        else:
            e = ref_train.elbo_loss(
                [recon_attrs[ix] for ix in range(attrs_combo.size) if attrs_combo[ix]],
                [attrs[ix] for ix in range(attrs_combo.size) if attrs_combo[ix]],
                mu, logvar, annealing_factor=BETA)
        train_loss += e; elbos.append(e); lat.append((mu, logvar))
    train_loss.backward()

    # ---- oracle ----
    terms = OS.celeba19_terms(combos)
    torch.manual_seed(noise_seed)
    noise = OS.draw_celeba19_noise(batch, d, terms)
    o_total, o_elbos, o_lat = OS.celeba19_step(oracle, image, attrs2d, terms, noise,
                                               lam_i, lam_a, BETA)
    o_total.backward()

    fx = {'image': image.numpy(), 'label': attrs2d.numpy(),
          'total': np.float64(train_loss.item()),
          'terms': np.array([e.item() for e in elbos], dtype=np.float64),
          'combos': np.asarray(combos, dtype=np.uint8),
          'logits_image_0': logits0}
    for c in range(len(terms)):
        fx['eps%d' % c] = noise['eps'][c].numpy()
        if noise['mask'][c] is not None:
            fx['mask%d' % c] = noise['mask'][c].numpy().astype(np.uint8)
        if c in (0, 1, 2, len(terms) - 1):
            fx['mu%d' % c] = lat[c][0].detach().numpy().copy()
            fx['logvar%d' % c] = lat[c][1].detach().numpy().copy()
            fx['z%d' % c] = o_lat[c][2].detach().numpy().copy()
        check(o_lat[c][0].detach(), lat[c][0].detach(), 'celeba19 mu%d' % c)
        check(o_lat[c][1].detach(), lat[c][1].detach(), 'celeba19 logvar%d' % c)
    fx.update(grad_digest(ref))
    fx.update(bn_stats(ref))
    check(o_total.item(), train_loss.item(), 'celeba19 total')
    check([e.item() for e in o_elbos], fx['terms'], 'celeba19 terms')
    og = grad_digest(oracle)
    for k in og:
        check(og[k], fx[k], 'celeba19 ' + k, rtol=1e-4, atol=1e-7)
    ob = bn_stats(oracle)
    for k in ob:
        check(ob[k], fx[k], 'celeba19 ' + k)
    meta = dict(exp=exp, batch=batch, n_latents=d, weight_seed=WEIGHT_SEED,
                noise_seed=noise_seed, input_seed=1234, lambda_image=lam_i,
                lambda_label=lam_a, beta=BETA, approx_m=approx_m, combo_seed=4321)
    return fx, meta


def save(name, fx, meta):
    fx = dict(fx)
    fx['meta'] = np.array(repr(meta))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **fx)
    print('%-28s %7.1f KB  total=%.6f' % (name, os.path.getsize(path) / 1024., fx['total']))


def main():
    torch.set_num_threads(4)
    for exp, batches in (('mnist', (4, 8)), ('fashionmnist', (4, 8)), ('celeba', (4, 8))):
        for b in batches:
            fx, meta = run_bimodal(exp, b, noise_seed=1000 + b)
            if b != 4:
                fx.pop('image')  # regenerate from input_seed (kept for B=4 as a cross-check)
            save('%s_b%d' % (exp, b), fx, meta)
    for b in (4, 8):            # B = 4 and B = 8 per experiment
        fx, meta = run_celeba19(b, noise_seed=1000 + b)
        if b != 4:
            fx.pop('image', None)   # regenerated from input_seed by the tests (kept for B = 4 as a cross-check)
        save('celeba19_b%d' % b, fx, meta)


if __name__ == '__main__':
    main()

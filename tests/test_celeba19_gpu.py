"""GPU parity of the 19-modality step (celeba19/train.py:236-316): fused engine vs the golden
fixture captured from the real reference and vs the live oracle; reference-surface calls."""
import numpy as np
import pytest
import torch

import mvae_amd
from mvae_amd.engine import Celeba19Step, sample_subsets
from oracle import models as OM, steps as OS
from test_engine_gpu import build_pair, check_bn_vs, check_grads_vs_golden, check_grads_vs_oracle
from util import assert_close, golden_noise, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('batch', [4, 8])
def test_fused_step_matches_reference_golden(golden_dir, batch):
    fx, meta = load_golden(golden_dir, 'celeba19_b%d' % batch)
    _, model, d = build_pair('celeba19', meta['weight_seed'])
    B = meta['batch']
    image, attrs = OS.synthetic_batch('celeba19', B, meta['input_seed'])
    combos = fx['combos'].astype(bool)
    T = 20 + meta['approx_m']
    eng = Celeba19Step(model, B, meta['lambda_image'], meta['lambda_label'], approx_m=meta['approx_m'])
    elbo = eng.step(image.to(DEV), attrs.to(DEV), meta['beta'], noise=golden_noise(fx, T), combos=combos).cpu()
    assert_close(elbo[:T], fx['terms'], 'ELBO terms')
    assert_close(elbo[T].item(), fx['total'], 'total loss')
    mu, lv, z = eng.last_latents
    for c in (0, 1, 2, T - 1):
        assert_close(mu[c], fx['mu%d' % c], 'mu%d' % c)
        assert_close(lv[c], fx['logvar%d' % c], 'logvar%d' % c)
        assert_close(z[c], fx['z%d' % c], 'z%d' % c)
    check_grads_vs_golden(model, fx)
    check_bn_vs(model, {k[3:]: v for k, v in fx.items() if k.startswith('bn/')})


@pytest.mark.parametrize('approx_m,batch,with_image', [(1, 6, True), (1, 6, False), (3, 4, None)])
def test_fused_step_matches_live_oracle(approx_m, batch, with_image):
    oracle, model, d = build_pair('celeba19', weight_seed=23)
    image, attrs = OS.synthetic_batch('celeba19', batch, seed=81)
    rng = np.random.RandomState(5 + approx_m)
    combos = sample_subsets(rng, 19, approx_m)
    if with_image is not None:
        combos[:, 0] = with_image
        if combos[0].sum() < 2:
            combos[0, 1:3] = True
    terms = OS.celeba19_terms(combos)
    torch.manual_seed(9)
    noise = OS.draw_celeba19_noise(batch, d, terms)
    lam_i, lam_a, beta = 1.0, 10.0, 0.3
    total, elbos, lat = OS.celeba19_step(oracle, image, attrs, terms, noise, lam_i, lam_a, beta)
    total.backward()
    eng = Celeba19Step(model, batch, lam_i, lam_a, approx_m=approx_m)
    elbo = eng.step(image.to(DEV), attrs.to(DEV), beta, noise=noise, combos=combos).cpu()
    T = len(terms)
    assert_close(elbo[:T], torch.stack(elbos).detach(), 'ELBO terms')
    assert_close(elbo[T], total.detach(), 'total')
    worst = check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())
    print('celeba19 M=%d worst gradient rel err %.2e' % (approx_m, worst))


def test_fused_step_matches_live_oracle_at_baseline_batch():
    """BASELINE.json configs[4]: batch 256 per GPU, approx-m 1, lambda_attrs 10; subsets drawn the way
    the reference's train loop draws them (same generator calls), 1e-4 on terms and every gradient,
    1e-5 on the BatchNorm running statistics after the 21 image-decoder / 2-3 trunk updates."""
    batch, approx_m = 256, 1
    oracle, model, d = build_pair('celeba19', weight_seed=41)
    image, attrs = OS.synthetic_batch('celeba19', batch, seed=93)
    combos = sample_subsets(np.random.RandomState(2024), 19, approx_m)
    terms = OS.celeba19_terms(combos)
    torch.manual_seed(11)
    noise = OS.draw_celeba19_noise(batch, d, terms)
    total, elbos, lat = OS.celeba19_step(oracle, image, attrs, terms, noise, 1.0, 10.0, 0.5)
    total.backward()
    eng = Celeba19Step(model, batch, 1.0, 10.0, approx_m=approx_m)
    elbo = eng.step(image.to(DEV), attrs.to(DEV), 0.5, noise=noise, combos=combos).cpu()
    T = len(terms)
    assert_close(elbo[:T], torch.stack(elbos).detach(), 'ELBO terms')
    assert_close(elbo[T], total.detach(), 'total')
    worst = check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())
    print('celeba19 B=256 M=1 (BASELINE size) worst gradient rel err %.2e' % worst)


def test_step_tables_match_a_plain_loop():
    """The per-step device tables (PoE masks, loss coefficients, BatchNorm update count) are filled
    vectorised into one pinned block; check them against the obvious per-entry loop
    (celeba19/train.py:265-302: which lambda each term uses)."""
    _, model, d = build_pair('celeba19', weight_seed=43)
    B, M = 4, 3
    eng = Celeba19Step(model, B, 2.0, 7.0, approx_m=M)
    combos = sample_subsets(np.random.RandomState(8), 19, M)
    combos[1, 0] = True; combos[2, 0] = False
    eng.set_terms(combos, commit=False)
    eng.set_coefficients(0.25)
    torch.cuda.synchronize()
    n_img, S, T = eng.n_img, eng.S, eng.T
    masks = [1 | (((1 << 18) - 1) << n_img), 1 << 1] + [1 << (n_img + i) for i in range(18)]
    coef_img = [2.0 / B, 2.0 / B] + [0.0] * 18
    for j in range(M):
        bits = (1 << (2 + j)) if combos[j, 0] else 0
        for i in range(18):
            if combos[j, 1 + i]:
                bits |= 1 << (n_img + i)
        masks.append(bits)
        coef_img.append(1.0 / B if combos[j, 0] else 0.0)
    assert eng.masks_dev.cpu().tolist() == masks
    assert eng.nimg_dev.item() == 2 + int(combos[:, 0].sum())
    np.testing.assert_allclose(eng.coef[0].cpu().numpy(), np.array(coef_img, dtype=np.float32))
    np.testing.assert_allclose(eng.coef[2].cpu().numpy(), np.full(T, 0.25 / B, dtype=np.float32))
    ca = eng.coef_attr.cpu().numpy().reshape(18, S)
    for i in range(18):
        expect = [7.0 / B] + [1.0 / B if combos[j, 1 + i] else 0.0 for j in range(M)] + [1.0 / B]
        np.testing.assert_allclose(ca[i], np.array(expect, dtype=np.float32))
    # the ring: many refreshes without a sync leave the LAST values on the device
    for k in range(9):
        eng.set_coefficients(0.1 * k)
    torch.cuda.synchronize()
    np.testing.assert_allclose(eng.coef[2].cpu().numpy(), np.full(T, 0.8 / B, dtype=np.float32), rtol=1e-6)


def test_module_surface_two_terms():
    """model(image, attrs) + elbo_loss on lists, as celeba19/train.py:264-283 calls them."""
    import mvae_amd.functional as MF
    from oracle import functional as OF
    oracle, model, d = build_pair('celeba19', weight_seed=29)
    B = 5
    image, attrs2d = OS.synthetic_batch('celeba19', B, seed=83)
    attrs = [attrs2d[:, i] for i in range(18)]
    g = torch.Generator().manual_seed(3)
    eps = [torch.randn(B, d, generator=g) for _ in range(2)]
    mask = (torch.rand(B, 512, generator=g) < 0.9).float()
    ri, ra, mu, lv, _ = oracle(image, attrs, eps=eps[0], dropout_mask=mask)
    e1 = OF.elbo_loss_multi([ri] + ra, [image] + attrs, mu, lv, 1.0, 10.0, 0.5)
    only3 = [attrs[k] if k == 3 else None for k in range(18)]
    _, ra2, mu2, lv2, _ = oracle(None, only3, eps=eps[1])
    e2 = OF.elbo_loss_multi([ra2[3]], [attrs[3]], mu2, lv2, annealing_factor=0.5)
    (e1 + e2).backward()

    img = image.to(DEV); at = [a.to(DEV) for a in attrs]
    model.zero_grad()
    hi, ha, hmu, hlv = model(img, at, eps=eps[0].to(DEV), dropout_mask=mask.to(DEV))
    h1 = MF.elbo_loss_multi([hi] + ha, [img] + at, hmu, hlv, lambda_image=1.0, lambda_attrs=10.0,
                            annealing_factor=0.5)
    _, ha2, hmu2, hlv2 = model(attrs=[at[k] if k == 3 else None for k in range(18)], eps=eps[1].to(DEV))
    h2 = MF.elbo_loss_multi([ha2[3]], [at[3]], hmu2, hlv2, annealing_factor=0.5)
    (h1 + h2).backward()
    assert_close(torch.stack([h1, h2]).detach(), torch.stack([e1, e2]).detach(), 'ELBO terms')
    assert_close(hmu, mu.detach(), 'mu')
    og = dict(oracle.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values() if p.grad is not None)
    for name, p in model.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, name
            continue
        scale = max(ref.abs().max().item(), 1e-30)      # celeba19 has no mathematically-zero gradients: own magnitude
        err = (p.grad.cpu() - ref).abs().max().item() / scale
        assert err <= 1e-4, 'grad %s: %.3e' % (name, err)


def test_graph_replay_with_changing_subsets():
    from mvae_amd.optim import FusedAdam
    _, model, d = build_pair('celeba19', weight_seed=31)
    B = 4
    eng = Celeba19Step(model, B, 1.0, 10.0, approx_m=1)
    opt = FusedAdam(model.parameters(), lr=1e-4)
    image, attrs = OS.synthetic_batch('celeba19', B, seed=85)
    eng.capture(opt, image.shape[1:], attrs)
    before = model.arena.flat.clone()
    losses = []
    for i in range(3):
        losses.append(eng.replay(image.to(DEV), attrs.to(DEV), 0.5)[-1].item())
    assert all(np.isfinite(losses))
    assert not torch.equal(before, model.arena.flat)


def test_constructing_an_engine_draws_nothing_from_the_subset_generator():
    """with ``rng = numpy.random`` (celeba19/train.py's global generator) every engine construction --
    also the ragged-last-batch engines built mid-epoch -- used to consume one ``sample_combinations`` draw the
    reference never makes."""
    _, model, d = build_pair('celeba19', weight_seed=47)
    rng = np.random.RandomState(99)
    before = rng.get_state()[1].copy(), rng.get_state()[2]
    eng = Celeba19Step(model, 4, 1.0, 10.0, approx_m=2, rng=rng)
    after = rng.get_state()[1], rng.get_state()[2]
    assert np.array_equal(before[0], after[0]) and before[1] == after[1]
    assert eng.combos.shape == (2, 19) and (eng.combos.sum(axis=1) == 2).all() and eng.combos[:, 0].all()
    # the first STEP takes the generator's first draw -- the reference's first sample_combinations call
    expect = sample_subsets(np.random.RandomState(99), 19, 2)
    image, attrs = OS.synthetic_batch('celeba19', 4, seed=86)
    eng.step(image.to(DEV), attrs.to(DEV), 0.5)
    assert np.array_equal(eng.combos, expect)


def test_loss_bearing_bn_stats_mode_keeps_elbo_and_gradients():
    """SURVEY Appendix B-4 decided explicitly (``celeba19/train.py --bn-stats loss-bearing`` -> ``faithful_bn_stats=False``):
    the 18 attribute-only terms' image decodes -- whose output the reference never reads (celeba19/train.py:278-283)
    -- are skipped.  ELBO terms and every gradient still equal the reference's (1e-4); what changes, and is the
    documented divergence, is the image decoder's BatchNorm state: 2 + M running-statistics updates instead of
    20 + M, so running_mean / running_var / num_batches_tracked differ; the image ENCODER's state does not."""
    approx_m, batch = 1, 6
    oracle, model, d = build_pair('celeba19', weight_seed=29)
    image, attrs = OS.synthetic_batch('celeba19', batch, seed=83)
    combos = sample_subsets(np.random.RandomState(11), 19, approx_m)
    terms = OS.celeba19_terms(combos)
    torch.manual_seed(10)
    noise = OS.draw_celeba19_noise(batch, d, terms)
    total, elbos, _ = OS.celeba19_step(oracle, image, attrs, terms, noise, 1.0, 10.0, 0.3)
    total.backward()
    eng = Celeba19Step(model, batch, 1.0, 10.0, approx_m=approx_m, faithful_bn_stats=False)
    elbo = eng.step(image.to(DEV), attrs.to(DEV), 0.3, noise=noise, combos=combos).cpu()
    T = len(terms)
    assert_close(elbo[:T], torch.stack(elbos).detach(), 'ELBO terms')
    assert_close(elbo[T], total.detach(), 'total')
    check_grads_vs_oracle(model, oracle)
    ref = oracle.state_dict()
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    enc = [k for k in ref if k.startswith('image_encoder') and 'running_' in k]
    dec = [k for k in ref if k.startswith('image_decoder') and 'running_mean' in k]
    assert enc and dec
    for k in enc:
        assert_close(got[k], ref[k], k, tol=1e-5)
    for k in dec:       # fewer momentum updates: visibly not the reference's state
        assert (got[k] - ref[k]).abs().max().item() > 1e-3 * ref[k].abs().max().item(), k
    nbt = [k for k in ref if k.startswith('image_decoder') and k.endswith('num_batches_tracked')]
    n_img_terms = 2 + int(combos[:, 0].sum())
    for k in nbt:
        assert int(ref[k]) == 20 + approx_m and int(got[k]) == n_img_terms, (k, int(ref[k]), int(got[k]))

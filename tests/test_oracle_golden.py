"""CPU: the oracle restatement reproduces the fixtures captured from the real reference
(tests/golden/make_golden.py).  This is what pins the oracle -- the reference has no tests of
its own."""
import numpy as np
import pytest
import torch

from oracle import models as OM, steps as OS
from util import assert_close, golden_noise, load_golden

BIMODAL = [('mnist', 4), ('mnist', 8), ('fashionmnist', 4), ('fashionmnist', 8), ('celeba', 4), ('celeba', 8)]


def _build(exp, meta):
    cls, d = OM.MODELS[exp]
    assert d == meta['n_latents']
    model = OM.fill_parameters(cls(d), meta['weight_seed'])
    model.train()
    return model


def _check_grads(model, fx):
    for name, p in model.named_parameters():
        g = p.grad.reshape(-1)
        assert_close(g.double().norm().item(), fx['gnorm/' + name], 'grad norm ' + name)
        ref = fx['ghead/' + name]
        scale = max(float(np.abs(ref).max()), float(fx['gnorm/' + name]) / max(g.numel(), 1) ** 0.5, 1e-12)
        err = np.abs(g[:8].numpy() - ref).max() / scale
        assert err <= 1e-4, 'grad head %s: %.3e' % (name, err)


def _check_bn(model, fx):
    for k, v in model.state_dict().items():
        if 'running_' in k or 'num_batches' in k:
            assert_close(v.double(), fx['bn/' + k].astype(np.float64), 'bn ' + k, tol=1e-5)


@pytest.mark.parametrize('exp,batch', BIMODAL)
def test_bimodal_step_matches_reference(golden_dir, exp, batch):
    torch.set_num_threads(4)
    fx, meta = load_golden(golden_dir, '%s_b%d' % (exp, batch))
    model = _build(exp, meta)
    image, label = OS.synthetic_batch(exp, batch, meta['input_seed'])
    if 'image' in fx:
        assert np.array_equal(image.numpy(), fx['image'])
    assert np.array_equal(label.numpy(), fx['label'])
    noise = golden_noise(fx, 3)
    total, terms, lat = OS.bimodal_step(model, exp, image, label, noise, meta['lambda_image'],
                                        meta['lambda_label'], meta['beta'])
    total.backward()
    assert_close(total.item(), fx['total'], 'total', tol=2e-6)
    assert_close([t.item() for t in terms], fx['terms'], 'terms', tol=2e-6)
    for c in range(3):
        assert_close(lat[c][0], fx['mu%d' % c], 'mu%d' % c, tol=1e-5)
        assert_close(lat[c][1], fx['logvar%d' % c], 'logvar%d' % c, tol=1e-5)
        assert_close(lat[c][2], fx['z%d' % c], 'z%d' % c, tol=1e-5)
    _check_grads(model, fx)
    _check_bn(model, fx)


def test_noise_replay_matches_global_generator(golden_dir):
    """draw_bimodal_noise under manual_seed(noise_seed) reproduces the recorded draws."""
    fx, meta = load_golden(golden_dir, 'celeba_b4')
    torch.manual_seed(meta['noise_seed'])
    noise = OS.draw_bimodal_noise(meta['batch'], meta['n_latents'], has_dropout=True)
    for c in range(3):
        assert np.array_equal(noise['eps'][c].numpy(), fx['eps%d' % c])
    assert np.array_equal(noise['mask'][0].numpy().astype(np.uint8), fx['mask0'])
    assert np.array_equal(noise['mask'][1].numpy().astype(np.uint8), fx['mask1'])
    assert noise['mask'][2] is None


@pytest.mark.parametrize('batch', [4, 8])
def test_celeba19_step_matches_reference(golden_dir, batch):
    torch.set_num_threads(4)
    fx, meta = load_golden(golden_dir, 'celeba19_b%d' % batch)
    assert meta['batch'] == batch
    model = _build('celeba19', meta)
    image, attrs = OS.synthetic_batch('celeba19', meta['batch'], meta['input_seed'])
    if 'image' in fx:
        assert np.array_equal(image.numpy(), fx['image'])
    assert np.array_equal(attrs.numpy(), fx['label'])
    terms = OS.celeba19_terms(fx['combos'].astype(bool))
    assert len(terms) == 20 + meta['approx_m']
    noise = golden_noise(fx, len(terms))
    total, elbos, lat = OS.celeba19_step(model, image, attrs, terms, noise, meta['lambda_image'],
                                         meta['lambda_label'], meta['beta'])
    total.backward()
    assert_close(total.item(), fx['total'], 'total', tol=2e-6)
    assert_close([e.item() for e in elbos], fx['terms'], 'terms', tol=2e-6)
    for c in (0, 1, 2, len(terms) - 1):
        assert_close(lat[c][0], fx['mu%d' % c], 'mu%d' % c, tol=1e-5)
        assert_close(lat[c][2], fx['z%d' % c], 'z%d' % c, tol=1e-5)
    _check_grads(model, fx)
    _check_bn(model, fx)


def test_celeba19_combination_pool_and_sampling():
    pool = OS.enumerate_combinations(6)
    # all subsets of sizes 2..5 of 6 modalities: C(6,2)+C(6,3)+C(6,4)+C(6,5) = 15+20+15+6
    assert pool.shape == (56, 6)
    sums = pool.sum(1)
    assert sums.min() == 2 and sums.max() == 5
    assert len({tuple(r) for r in pool}) == 56
    rng = np.random.RandomState(0)
    s = OS.sample_combinations(pool, size=7, rng=rng)
    assert s.shape == (7, 6) and s.dtype == bool
    assert ((s.sum(1) >= 2) & (s.sum(1) <= 5)).all()


def test_loss_helpers_error_behaviour():
    """Same ValueError as the reference on mismatched sizes (mnist/train.py:69-71,85-88)."""
    from oracle import functional as OF
    with pytest.raises(ValueError, match='Target size'):
        OF.binary_cross_entropy_with_logits(torch.zeros(4, 3), torch.zeros(4, 2))
    with pytest.raises(ValueError, match='Target size'):
        OF.cross_entropy(torch.zeros(4, 10), torch.zeros(3, dtype=torch.long))


# ----------------------------------------------------------------------------------------------------
# the reference's test(epoch) body in eval mode (tests/golden/make_eval_golden.py): reparametrize returns mu,
# Dropout off, BatchNorm on running statistics, elbo_loss at its default weights
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('exp', ['mnist', 'fashionmnist', 'celeba', 'celeba19'])
def test_eval_pass_matches_reference(golden_dir, exp):
    from oracle import functional as OF
    torch.set_num_threads(4)
    fx, meta = load_golden(golden_dir, 'eval_' + exp)
    model = _build(exp, meta)
    sd = model.state_dict()
    moved = 0
    for k in list(sd):
        if 'running_' in k or 'num_batches' in k:
            sd[k] = torch.from_numpy(np.asarray(fx['bn/' + k])).to(sd[k].dtype)
            moved += 1
    model.load_state_dict(sd)
    assert (moved > 0) == (exp in ('celeba', 'celeba19'))
    model.eval()
    image, label = OS.synthetic_batch(exp, meta['batch'], meta['input_seed'])
    assert np.array_equal(label.numpy(), fx['label'])
    with torch.no_grad():
        if exp == 'celeba19':
            attrs = [label[:, i] for i in range(label.shape[1])]
            ri, ra, mu, lv, z = model(image, attrs)
            calls = [(ri, torch.stack(ra, dim=1), mu, lv, z)]
            terms = [OF.elbo_loss_multi([ri] + ra, [image] + attrs, mu, lv)]
        else:
            elbo = OF.elbo_loss_attrs if exp == 'celeba' else OF.elbo_loss_label
            lam = (meta['lambda_image'], meta['lambda_label'])      # celeba/train.py:238-243 passes the CLI's; 1, 1 elsewhere
            calls = [model(image, label), model(image, None), model(None, label)]
            terms = [elbo(calls[0][0], image, calls[0][1], label, calls[0][2], calls[0][3], *lam),
                     elbo(calls[1][0], image, None, None, calls[1][2], calls[1][3], *lam),
                     elbo(None, None, calls[2][1], label, calls[2][2], calls[2][3], *lam)]
    assert_close([t.item() for t in terms], fx['terms'], 'eval terms', tol=2e-6)
    assert_close(sum(t.item() for t in terms), fx['total'], 'eval total', tol=2e-6)
    for c, (ri, rl, mu, lv, z) in enumerate(calls):
        assert_close(mu, fx['mu%d' % c], 'eval mu%d' % c, tol=1e-5)
        assert_close(lv, fx['logvar%d' % c], 'eval logvar%d' % c, tol=1e-5)
        assert torch.equal(z, mu), 'eval-mode reparametrize returns mu'
        assert_close(ri[0].reshape(-1)[:256], fx['logits_image_0_%d' % c], 'eval image logits %d' % c, tol=1e-5)
        assert_close(rl, fx['logits_label_%d' % c], 'eval label logits %d' % c, tol=1e-5)


def test_multimnist_text_oracle_reproduces_the_reference_golden(golden_dir):
    """SURVEY 8f-4's GRU stacks: oracle/multimnist.py against the fixture captured from the imported
    multimnist/model.py + train.py (tests/golden/make_multimnist_golden.py)."""
    import numpy as np
    from oracle import multimnist as OMM
    fx, meta = load_golden(golden_dir, 'multimnist_text')
    enc = OM.fill_parameters(OMM.TextEncoder(meta['n_latents']), meta['enc_seed']).train()
    dec = OM.fill_parameters(OMM.TextDecoder(meta['n_latents']), meta['dec_seed']).train()
    text = torch.from_numpy(fx['text'])
    assert torch.equal(text, OMM.synthetic_text(meta['batch'], meta['text_seed']))
    masks = [torch.from_numpy(fx['mask%d' % i]).float() for i in range(OMM.MAX_LENGTH)]
    torch.manual_seed(meta['noise_seed'])
    assert all(torch.equal(a, b) for a, b in zip(masks, OMM.draw_decoder_masks(meta['batch'])))
    mu, logvar = enc(text)
    words, fed = dec(mu + 0.5 * logvar, dropout_masks=masks)
    loss = OMM.text_loss_rows(words, text).mean() + 0.1 * (mu.pow(2) + logvar.pow(2)).mean()
    loss.backward()
    assert_close(mu, fx['mu'], 'mu', tol=1e-5); assert_close(words, fx['words'], 'words', tol=1e-5)
    assert np.array_equal(fed.numpy(), fx['fed'])
    assert_close(loss.item(), fx['loss'], 'loss', tol=1e-5)
    for prefix, mod in (('text_encoder', enc), ('text_decoder', dec)):
        for name, p in mod.named_parameters():
            ref = float(fx['gnorm/%s.%s' % (prefix, name)])
            assert abs(p.grad.double().norm().item() - ref) <= 2e-5 * max(ref, 1e-30), name
    dec.eval()
    with torch.no_grad():
        assert_close(dec((mu + 0.5 * logvar).detach())[0], fx['eval_words'], 'eval words', tol=1e-5)

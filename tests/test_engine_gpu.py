"""GPU parity proper: the HIP train step (fused engine and the reference-surface module path)
against (i) the golden fixtures captured from the real reference and (ii) the live oracle on
fresh seeds at other batch sizes.  Bar: 1e-4 relative on ELBO terms, latents and every
gradient (north_star); BatchNorm running statistics 1e-5."""
import numpy as np
import pytest
import torch

import mvae_amd
from mvae_amd.engine import BimodalStep
from mvae_amd.optim import FusedAdam
from oracle import models as OM, steps as OS
from util import (REL_TOL, assert_close, assert_zero_grad, golden_noise, is_zero_grad, load_golden, note_redraws,
                  zero_grad_weight)

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build_pair(kind, weight_seed):
    cls, d = OM.MODELS[kind]
    oracle = OM.fill_parameters(cls(d), weight_seed)
    oracle.train()
    model = getattr(mvae_amd, kind).model.MVAE(d)
    model.load_state_dict(oracle.state_dict())
    model.to(DEV).train()
    model.finalize()
    return oracle, model, d


def model_kind(model):
    return type(model).__module__.split('.')[-2]


def check_grads_vs_golden(model, fx):
    kind = model_kind(model)
    params = dict(model.named_parameters())
    for name, p in params.items():
        assert p.grad is not None, 'no gradient for ' + name
        gv = p.grad.detach().reshape(-1).cpu()
        ref_norm = float(fx['gnorm/' + name])
        if is_zero_grad(kind, name):
            # exactly zero in exact arithmetic: each side must be round-off of the same layer's weight gradient
            wn = zero_grad_weight(name)
            assert_zero_grad(name, gv.abs().max().item(), params[wn].grad.abs().max().item(), 'HIP')
            # the fixture keeps norms, not maxima: |g|_2 <= sqrt(n) max|g| and max|g_w| >= |g_w|_2 / sqrt(n_w)
            w_rms = float(fx['gnorm/' + wn]) / params[wn].numel() ** 0.5
            assert_zero_grad(name, ref_norm / gv.numel() ** 0.5, w_rms * 10.0, 'reference fixture (rms)')
            continue
        err = abs(gv.double().norm().item() - ref_norm) / max(ref_norm, 1e-30)
        assert err <= REL_TOL, 'grad norm %s: %.3e' % (name, err)
        ref = fx['ghead/' + name]
        scale = max(float(np.abs(ref).max()), ref_norm / max(gv.numel(), 1) ** 0.5, 1e-30)
        err = np.abs(gv[:8].numpy() - ref).max() / scale
        assert err <= REL_TOL, 'grad head %s: %.3e' % (name, err)


def check_grads_vs_oracle(model, oracle, tol=REL_TOL):
    kind = model_kind(model)
    og = dict(oracle.named_parameters())
    params = dict(model.named_parameters())
    worst, bad = 0.0, []
    for name, p in params.items():
        assert p.grad is not None, 'no gradient for ' + name
        ref = og[name].grad
        if is_zero_grad(kind, name):
            wn = zero_grad_weight(name)
            assert_zero_grad(name, p.grad.abs().max().item(), params[wn].grad.abs().max().item(), 'HIP')
            assert_zero_grad(name, ref.abs().max().item(), og[wn].grad.abs().max().item(), 'oracle')
            continue
        scale = max(ref.abs().max().item(), 1e-30)
        err = (p.grad.detach().cpu() - ref).abs().max().item() / scale
        if err > tol:
            bad.append('%s %.3e' % (name, err))
        worst = max(worst, err)
    assert not bad, 'gradients beyond %.0e: %s' % (tol, '; '.join(bad))
    return worst


def hits_bce_jump(eng):
    """The reference's hand-written BCE (mnist/train.py:73-74) has a gradient JUMP at a logit of exactly 0
    (autograd gives 1 - t there, sigmoid(0) - t = 0.5 - t an ulp away).  A logit that
    is a few ulps of its bias from zero can round to 0.0 on one side and not on the other -- then ONE
    d loss / d logit differs by lambda/(2B) and every gradient behind it by ~1e-2, which says nothing about
    the kernels.  Parity cases whose HIP logits contain an exact zero are re-drawn with the next seed."""
    logits_img, logits_lbl = eng.recon_logits()      # a decoder whose loss rides its last Linear re-issues that launch
    return bool((logits_img == 0).any().item()) or (logits_lbl.dtype == torch.float32 and eng.model.LABEL_KIND != 'class'
                                                     and bool((logits_lbl == 0).any().item()))


def check_bn_vs(model, ref_sd, tol=1e-5):
    sd = model.state_dict()
    for k, v in ref_sd.items():
        if 'running_' in k or 'num_batches' in k:
            assert_close(sd[k].double(), torch.as_tensor(np.asarray(v)).double(), 'bn ' + k, tol=tol)


@pytest.mark.parametrize('kind,batch', [('mnist', 4), ('mnist', 8), ('fashionmnist', 4), ('fashionmnist', 8),
                                        ('celeba', 4), ('celeba', 8)])
def test_fused_step_matches_reference_goldens(golden_dir, kind, batch):
    fx, meta = load_golden(golden_dir, '%s_b%d' % (kind, batch))
    _, model, d = build_pair(kind, meta['weight_seed'])
    image, label = OS.synthetic_batch(kind, batch, meta['input_seed'])
    eng = BimodalStep(model, batch, meta['lambda_image'], meta['lambda_label'])
    elbo = eng.step(image.to(DEV), label.to(DEV), meta['beta'], noise=golden_noise(fx, 3))
    terms = eng.terms_in_reference_order(elbo).cpu()
    assert_close(terms[:3], fx['terms'], 'ELBO terms')
    assert_close(terms[3].item(), fx['total'], 'total loss')
    mu, lv, z = eng.last_latents
    for c in range(3):
        t = eng.ref_order.index(c)
        assert_close(mu[t], fx['mu%d' % c], 'mu%d' % c)
        assert_close(lv[t], fx['logvar%d' % c], 'logvar%d' % c)
        assert_close(z[t], fx['z%d' % c], 'z%d' % c)
    check_grads_vs_golden(model, fx)
    check_bn_vs(model, {k[3:]: v for k, v in fx.items() if k.startswith('bn/')})


# ragged sizes too: odd batches leave partial tiles / partial float4 groups in every kernel.  (CelebA
# at batch 2 is left out: BatchNorm1d over two samples maps every input to +-1, the gradient behind it is
# exactly zero in exact arithmetic, and what either implementation returns there is round-off.)
# ('mnist', 128) is BASELINE.json configs[0]'s exact batch (the reference's own CPU-runnable case)
@pytest.mark.parametrize('kind,batch', [('mnist', 96), ('fashionmnist', 40), ('celeba', 12), ('mnist', 1),
                                        ('mnist', 67), ('fashionmnist', 13), ('celeba', 5), ('celeba', 7),
                                        ('mnist', 128)])
def test_fused_step_matches_live_oracle(kind, batch):
    oracle, model, d = build_pair(kind, weight_seed=11)
    image, label = OS.synthetic_batch(kind, batch, seed=77)
    torch.manual_seed(5)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))
    lam_i, lam_l, beta = 1.0, (10.0 if kind == 'celeba' else 50.0), 0.37
    total, terms, lat = OS.bimodal_step(oracle, kind, image, label, noise, lam_i, lam_l, beta)
    total.backward()
    eng = BimodalStep(model, batch, lam_i, lam_l)
    elbo = eng.terms_in_reference_order(eng.step(image.to(DEV), label.to(DEV), beta, noise=noise)).cpu()
    assert_close(elbo[:3], torch.stack(terms).detach(), 'ELBO terms')
    assert_close(elbo[3], total.detach(), 'total')
    worst = check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())
    print('%s B=%d worst gradient rel err %.2e' % (kind, batch, worst))


# BASELINE.json's configurations at their full per-GPU batch sizes (configs[1..3]): the regime where the
# weight-gradient reductions are longest (CelebA B=256: 262,144 terms over up to 64 split partials) and
# fp32 error is largest.  Same bar: ELBO terms, every gradient 1e-4, BatchNorm running statistics 1e-5.
@pytest.mark.parametrize('kind,batch', [('mnist', 512), ('fashionmnist', 1024), ('celeba', 256)])
def test_fused_step_matches_live_oracle_at_baseline_batch(kind, batch):
    for attempt in range(6):
        oracle, model, d = build_pair(kind, weight_seed=37)
        image, label = OS.synthetic_batch(kind, batch, seed=91 + attempt)
        torch.manual_seed(7)
        noise = OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))
        lam_i, lam_l, beta = 1.0, (10.0 if kind == 'celeba' else 50.0), 0.5
        eng = BimodalStep(model, batch, lam_i, lam_l)
        elbo = eng.terms_in_reference_order(eng.step(image.to(DEV), label.to(DEV), beta, noise=noise)).cpu()
        total, terms, lat, recon = OS.bimodal_step(oracle, kind, image, label, noise, lam_i, lam_l, beta,
                                                   return_recon=True)
        # an exactly-zero logit on EITHER side sits on the BCE gradient's jump (see hits_bce_jump): re-draw, and say so
        bce_logits = [r[0] for r in recon if r[0] is not None]
        if kind == 'celeba':
            bce_logits += [r[1] for r in recon if r[1] is not None]
        if not (hits_bce_jump(eng) or any(bool((x == 0).any()) for x in bce_logits)):
            break
    else:
        pytest.fail('six consecutive draws with an exactly-zero logit')
    total.backward()
    assert_close(elbo[:3], torch.stack(terms).detach(), 'ELBO terms')
    assert_close(elbo[3], total.detach(), 'total')
    mu, lv, z = eng.last_latents
    for c in range(3):
        t = eng.ref_order.index(c)
        assert_close(mu[t], lat[c][0].detach(), 'mu%d' % c)
        assert_close(lv[t], lat[c][1].detach(), 'logvar%d' % c)
    worst = check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())
    print('%s B=%d (BASELINE size) worst gradient rel err %.2e; %d re-draw(s) for an exactly-zero logit'
          % (kind, batch, worst, attempt))
    # an exact zero among 10^5..10^7 fp32 logits is a rare accident of one draw; a kernel that MANUFACTURED zeros would
    # need re-draw after re-draw -- that must fail, not be retried away
    note_redraws('engine eager %s B=%d' % (kind, batch), attempt)
    assert attempt <= 1, '%d re-draws for exactly-zero logits' % attempt


@pytest.mark.parametrize('kind,batch', [('mnist', 24), ('fashionmnist', 9)])
def test_paired_decoder_launches_match_live_oracle(kind, batch, monkeypatch):
    """MVAE_PAIR=1: the decoders' shared leading Linear layers as one launch for both (opt-in path)."""
    monkeypatch.setenv('MVAE_PAIR', '1')
    oracle, model, d = build_pair(kind, weight_seed=13)
    image, label = OS.synthetic_batch(kind, batch, seed=78)
    torch.manual_seed(6)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=False)
    total, terms, _ = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, 50.0, 0.41)
    total.backward()
    eng = BimodalStep(model, batch, 1.0, 50.0)
    assert eng.pair_dec == (3 if kind == 'mnist' else 1)
    elbo = eng.terms_in_reference_order(eng.step(image.to(DEV), label.to(DEV), 0.41, noise=noise)).cpu()
    assert_close(elbo[:3], torch.stack(terms).detach(), 'ELBO terms')
    check_grads_vs_oracle(model, oracle)


@pytest.mark.parametrize('batch', [24, 512])
def test_paired_encoder_launches_match_live_oracle(batch, monkeypatch):
    """MVAE_PAIR_ENC=1: MNIST's two encoders share the 512 -> 512 layer and the head pair (mnist/model.py:76-78,117-119);
    as G = 2 launches on one stream the encoder phases have no fork / join (opt-in: measured 1 % slower, engine.py)."""
    monkeypatch.setenv('MVAE_PAIR_ENC', '1')
    oracle, model, d = build_pair('mnist', weight_seed=17)
    image, label = OS.synthetic_batch('mnist', batch, seed=79)
    torch.manual_seed(8)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=False)
    total, terms, _ = OS.bimodal_step(oracle, 'mnist', image, label, noise, 1.0, 50.0, 0.43)
    total.backward()
    eng = BimodalStep(model, batch, 1.0, 50.0)
    assert eng.pair_enc == 2
    elbo = eng.terms_in_reference_order(eng.step(image.to(DEV), label.to(DEV), 0.43, noise=noise)).cpu()
    assert_close(elbo[:3], torch.stack(terms).detach(), 'ELBO terms')
    assert_close(elbo[3], total.detach(), 'total')
    check_grads_vs_oracle(model, oracle)
    # fashionmnist / celeba have nothing to pair (conv trunk, BatchNorm)
    _, other, _ = build_pair('fashionmnist', weight_seed=17)
    assert BimodalStep(other, 8, 1.0, 50.0).pair_enc == 0


# ... and at BASELINE.json's per-GPU batches ("What's missing" 3): the reference-shaped loop on the drop-in modules
# is what bench.py's `module_surface` times, and 512 / 1024 / 256 rows are where its split reductions are longest
@pytest.mark.parametrize('kind,batch', [('mnist', 16), ('fashionmnist', 8), ('celeba', 6),
                                        ('mnist', 512), ('fashionmnist', 1024), ('celeba', 256)])
def test_module_surface_matches_live_oracle(kind, batch):
    """The reference's own call pattern: three model() calls, three elbo_loss calls, backward
    (mnist/train.py:200-218) on the drop-in nn.Module + functional surface."""
    import mvae_amd.functional as MF
    for attempt in range(3):
        oracle, model, d = build_pair(kind, weight_seed=13)
        image, label = OS.synthetic_batch(kind, batch, seed=78 + attempt)
        torch.manual_seed(6)
        noise = OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))
        lam_i, lam_l, beta = 1.0, 10.0, 0.5
        total, terms, lat, recon = OS.bimodal_step(oracle, kind, image, label, noise, lam_i, lam_l, beta, return_recon=True)
        # an exactly-zero logit sits on the reference BCE's gradient jump: only the large batches ever hit one
        logits = [r[0] for r in recon if r[0] is not None] + ([r[1] for r in recon if r[1] is not None] if kind == 'celeba' else [])
        if not any(bool((x == 0).any()) for x in logits):
            break
    note_redraws('module surface %s B=%d' % (kind, batch), attempt)
    assert attempt <= 1
    total.backward()

    img, lbl = image.to(DEV), label.to(DEV)
    eps = [e.to(DEV) for e in noise['eps']]
    model.zero_grad()
    if kind == 'celeba':
        elbo = MF.elbo_loss_attrs
        r1 = model(img, lbl, eps=eps[0], dropout_mask=noise['mask'][0].to(DEV))
        r2 = model(img, eps=eps[1], dropout_mask=noise['mask'][1].to(DEV))
        r3 = model(attrs=lbl, eps=eps[2])
        kw = dict(lambda_image=lam_i, lambda_attrs=lam_l, annealing_factor=beta)
    else:
        elbo = MF.elbo_loss_label
        r1 = model(img, lbl, eps=eps[0])
        r2 = model(img, eps=eps[1])
        r3 = model(text=lbl, eps=eps[2])
        kw = dict(lambda_image=lam_i, lambda_text=lam_l, annealing_factor=beta)
    joint = elbo(r1[0], img, r1[1], lbl, r1[2], r1[3], **kw)
    iloss = elbo(r2[0], img, None, None, r2[2], r2[3], **kw)
    lloss = elbo(None, None, r3[1], lbl, r3[2], r3[3], **kw)
    train_loss = joint + iloss + lloss
    train_loss.backward()
    assert_close(torch.stack([joint, iloss, lloss]).detach(), torch.stack(terms).detach(), 'ELBO terms')
    assert_close(r1[2], lat[0][0].detach(), 'mu (joint)')
    assert_close(r1[3], lat[0][1].detach(), 'logvar (joint)')
    check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())


def test_eval_mode_and_error_behaviour():
    import mvae_amd.functional as MF
    oracle, model, d = build_pair('celeba', weight_seed=17)
    image, label = OS.synthetic_batch('celeba', 5, seed=79)
    oracle.eval(); model.eval()
    with torch.no_grad():
        ri, ra, mu, lv, z = oracle(image, label)
        hi, ha, hmu, hlv = model(image.to(DEV), label.to(DEV))
    assert_close(hmu, mu, 'eval mu'); assert_close(hlv, lv, 'eval logvar')
    assert_close(hi, ri, 'eval image logits'); assert_close(ha, ra, 'eval attr logits')
    with pytest.raises(ValueError, match='Target size'):
        MF.binary_cross_entropy_with_logits(torch.zeros(4, 3, device=DEV), torch.zeros(4, 2, device=DEV))
    with pytest.raises(ValueError, match='Target size'):
        MF.cross_entropy(torch.zeros(4, 10, device=DEV), torch.zeros(3, dtype=torch.long, device=DEV))
    x = torch.randn(6, 10, device=DEV); y = torch.randint(0, 10, (6,), device=DEV)
    from oracle import functional as OF
    assert_close(MF.cross_entropy(x, y), OF.cross_entropy(x.cpu(), y.cpu()), 'cross_entropy matrix')
    t = torch.rand(6, 10, device=DEV)
    assert_close(MF.binary_cross_entropy_with_logits(x, t),
                 OF.binary_cross_entropy_with_logits(x.cpu(), t.cpu()), 'bce elementwise')


def test_training_trajectory_with_fused_adam_and_graph():
    """5 optimizer steps: eager engine + FusedAdam vs oracle + torch.optim.Adam on the same
    noise; then the same model under hipGraph replay keeps training (loss finite, parameters
    move, captured state restored exactly)."""
    kind, batch = 'mnist', 32
    oracle, model, d = build_pair(kind, weight_seed=19)
    opt_ref = torch.optim.Adam(oracle.parameters(), lr=1e-3)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    eng = BimodalStep(model, batch, 1.0, 50.0)
    for step in range(5):
        image, label = OS.synthetic_batch(kind, batch, seed=200 + step)
        torch.manual_seed(300 + step)
        noise = OS.draw_bimodal_noise(batch, d, has_dropout=False)
        opt_ref.zero_grad()
        total, _, _ = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, 50.0, 0.1 * (step + 1))
        total.backward(); opt_ref.step()
        elbo = eng.step(image.to(DEV), label.to(DEV), 0.1 * (step + 1), noise=noise)
        opt.step()
        assert_close(elbo[3], total.detach(), 'loss at step %d' % step)
    # Adam's update lr * m / (sqrt(v) + eps) is a sign function for |g| ~ eps: elements whose
    # gradient is round-off move by up to lr per step in either direction, so parameters are
    # compared on the size of the total movement (5 * lr), not at 1e-4
    for (n, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        err = (p.detach().cpu() - q.detach()).abs().max().item()
        assert err <= 0.2 * 5 * 1e-3, 'param after 5 steps %s: abs err %.3e' % (n, err)
    before = model.arena.flat.clone()
    image, label = OS.synthetic_batch(kind, batch, seed=400)
    eng.capture(opt, image.shape[1:], label)
    assert torch.equal(before, model.arena.flat), 'capture() must not change the parameters'
    losses = []
    for step in range(4):
        image, label = OS.synthetic_batch(kind, batch, seed=500 + step)
        losses.append(eng.replay(image.to(DEV), label.to(DEV), 0.5)[3].item())
    assert all(np.isfinite(losses)) and len(set(losses)) == 4
    assert not torch.equal(before, model.arena.flat)
    assert opt._step_dev.item() == 5 + 4


def test_captured_step_with_early_counter_equals_update_plus_counter_launch(monkeypatch):
    """The captured single-GPU step advances Adam's step counter at the start of the step on the side stream and
    updates in one launch; MVAE_EARLY_COUNTER=0 is the update + counter launch at the end.  Same parameters, bit
    for bit, and the same counter."""
    kind, batch = 'mnist', 32
    finals = []
    for early in ('1', '0'):
        monkeypatch.setenv('MVAE_EARLY_COUNTER', early)
        _, model, _ = build_pair(kind, weight_seed=23)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        eng = BimodalStep(model, batch, 1.0, 50.0)
        image, label = OS.synthetic_batch(kind, batch, seed=600)
        eng.capture(opt, image.shape[1:], label)
        for step in range(3):
            image, label = OS.synthetic_batch(kind, batch, seed=610 + step)
            eng.replay(image.to(DEV), label.to(DEV), 0.5)
        torch.cuda.synchronize()
        assert opt._step_dev.item() == 3
        finals.append(model.arena.flat.clone())
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize('kind,batch', [('mnist', 24), ('celeba', 6)])
def test_decoders_updated_early_is_the_same_update(kind, batch, monkeypatch):
    """MVAE_SPLIT_ADAM=1: the captured step runs Adam on the decoders' arena range as soon as their weight gradients are
    final (main stream, beside the encoders' backward) and on the encoders' range at the end -- element for element the
    arithmetic of the one arena-wide launch: parameters, moments and counter equal to the last bit after 3 replays."""
    finals = []
    for split in ('1', '0'):
        monkeypatch.setenv('MVAE_SPLIT_ADAM', split)
        _, model, _ = build_pair(kind, weight_seed=31)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        eng = BimodalStep(model, batch, 1.0, 10.0 if kind == 'celeba' else 50.0, seed=9)
        image, label = OS.synthetic_batch(kind, batch, seed=720)
        eng.capture(opt, image.shape[1:], label)
        for step in range(3):
            image, label = OS.synthetic_batch(kind, batch, seed=730 + step)
            eng.replay(image.to(DEV), label.to(DEV), 0.5)
        torch.cuda.synchronize()
        assert opt._step_dev.item() == 3
        finals.append((model.arena.flat.clone(), opt._m.clone(), opt._v.clone()))
    for a, b, what in zip(finals[0], finals[1], ('parameters', 'exp_avg', 'exp_avg_sq')):
        assert torch.equal(a, b), what


@pytest.mark.parametrize('kind,batch', [('mnist', 24), ('mnist', 512), ('fashionmnist', 8)])
def test_weight_gradient_launches_that_update_their_parameters(kind, batch, monkeypatch):
    """MVAE_FUSE_ADAM=1: the Linear weight-gradient batches of the captured single-GPU step run optimizer.step() on
    their own outputs (FashionMNIST's conv / BatchNorm parameters stay with the launch at the end of the chain).  Same
    parameters, moments and counter, bit for bit, as the arena-wide update -- and on MNIST nothing is left for a launch
    at the end.  (Off by default: measured slower, profiles/r04_fuse_adam_ab.txt.)  Since round 5 the plain batches run on
    another tile code than the fused ones (64-wide wave tiles, a different summation tree): the two runs agree to fp32
    round-off, not bit for bit -- the bit-exact statement lives in tests/test_kernels_gpu.py (Adam on the gradients the
    fused launch produced)."""
    finals = []
    for fuse in ('1', '0'):
        monkeypatch.setenv('MVAE_FUSE_ADAM', fuse)
        _, model, _ = build_pair(kind, weight_seed=29)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        eng = BimodalStep(model, batch, 1.0, 50.0)
        assert eng.fuse_adam == (fuse == '1')
        image, label = OS.synthetic_batch(kind, batch, seed=700)
        eng.capture(opt, image.shape[1:], label)
        if fuse == '1':
            rest = opt.fusion().rest()
            covered = sum(b - a for a, b in opt.fusion().covered)
            assert covered > 0
            if kind == 'mnist':
                assert rest == [], rest
            else:
                assert rest and sum(b - a for a, b in rest) + covered <= model.arena.numel
        for step in range(3):
            image, label = OS.synthetic_batch(kind, batch, seed=710 + step)
            eng.replay(image.to(DEV), label.to(DEV), 0.5)
        torch.cuda.synchronize()
        assert opt._step_dev.item() == 3
        finals.append((model.arena.flat.clone(), opt._m.clone(), opt._v.clone(), model.arena.grad.clone()))
    lr, steps = 1e-3, 3
    for a, b, what in zip(finals[0], finals[1], ('parameters', 'exp_avg', 'exp_avg_sq', 'gradients')):
        if what == 'parameters':
            # an Adam step is lr * m / (sqrt(v) + eps): round-off in g moves it by O(1e-6 lr), up to a few % of lr where |g| ~ eps
            assert (a - b).abs().max().item() <= 0.05 * lr * steps, what
        else:
            scale = max(b.abs().max().item(), 1e-30)
            assert (a - b).abs().max().item() <= 1e-5 * scale, what

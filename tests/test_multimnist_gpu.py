"""GPU parity of the MultiMNIST recurrent text stacks (multimnist/model.py:145-235): the HIP
TextEncoder / TextDecoder against the golden fixture captured from the real reference and against the live oracle,
forward and backward, training mode (inter-layer dropout masks replayed) and eval mode; and the K16 kernels one by
one.  Bar: 1e-4 relative (outputs, loss, every gradient); the fed-back characters bit-exact."""
import numpy as np
import pytest
import torch

import mvae_amd  # noqa: F401
from mvae_amd import kernels as K
from mvae_amd.multimnist import model as MM
from oracle import models as OM, multimnist as OMM
from util import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build(n_latents, enc_seed, dec_seed, train=True):
    o_enc = OM.fill_parameters(OMM.TextEncoder(n_latents), enc_seed)
    o_dec = OM.fill_parameters(OMM.TextDecoder(n_latents), dec_seed)
    enc = MM.TextEncoder(n_latents, MM.n_characters, n_hiddens=200, bidirectional=True)
    dec = MM.TextDecoder(n_latents, MM.n_characters, n_hiddens=200)
    enc.load_state_dict(o_enc.state_dict()); dec.load_state_dict(o_dec.state_dict())
    enc.to(DEV); dec.to(DEV)
    for m in (o_enc, o_dec, enc, dec):
        m.train(train)
    return o_enc, o_dec, enc, dec


def hip_objective(enc, dec, text, masks):
    import mvae_amd.functional as MF
    mu, logvar = enc(text)
    z = mu + 0.5 * logvar
    words = dec(z, dropout_masks=masks)
    B, L, Kc = words.shape
    rows = MF.cross_entropy(words.reshape(-1, Kc), text.reshape(-1)).sum(dim=1).view(B, L).sum(dim=1)
    loss = rows.mean() + 0.1 * (mu.pow(2) + logvar.pow(2)).mean()
    return loss, mu, logvar, words


def oracle_objective(enc, dec, text, masks):
    mu, logvar = enc(text)
    z = mu + 0.5 * logvar
    words, fed = dec(z, dropout_masks=masks)
    loss = OMM.text_loss_rows(words, text).mean() + 0.1 * (mu.pow(2) + logvar.pow(2)).mean()
    return loss, mu, logvar, words, fed


def test_text_stacks_match_reference_golden(golden_dir):
    fx, meta = load_golden(golden_dir, 'multimnist_text')
    _, _, enc, dec = build(meta['n_latents'], meta['enc_seed'], meta['dec_seed'])
    text = torch.from_numpy(fx['text']).to(DEV)
    masks = [torch.from_numpy(fx['mask%d' % i]).float() for i in range(4)]
    loss, mu, logvar, words = hip_objective(enc, dec, text, masks)
    loss.backward()
    assert_close(mu, fx['mu'], 'mu'); assert_close(logvar, fx['logvar'], 'logvar')
    assert_close(words, fx['words'], 'words')
    assert np.array_equal(dec.last_fed.cpu().numpy(), fx['fed']), 'fed-back characters'
    assert_close(loss.item(), fx['loss'], 'loss')
    for prefix, mod in (('text_encoder', enc), ('text_decoder', dec)):
        for name, p in mod.named_parameters():
            g = p.grad.detach().reshape(-1).cpu()
            ref_norm = float(fx['gnorm/%s.%s' % (prefix, name)])
            assert abs(g.double().norm().item() - ref_norm) <= 1e-4 * max(ref_norm, 1e-30), 'grad norm ' + name
            ref = fx['ghead/%s.%s' % (prefix, name)]
            scale = max(float(np.abs(ref).max()), ref_norm / max(g.numel(), 1) ** 0.5, 1e-30)
            assert np.abs(g[:8].numpy() - ref).max() <= 1e-4 * scale, 'grad head ' + name
    dec.eval()
    with torch.no_grad():
        assert_close(dec((mu + 0.5 * logvar).detach()), fx['eval_words'], 'eval words')


@pytest.mark.parametrize('batch,n_latents,train', [(1, 64, True), (37, 64, True), (256, 100, True), (19, 64, False)])
def test_text_stacks_match_live_oracle(batch, n_latents, train):
    o_enc, o_dec, enc, dec = build(n_latents, 21, 22, train)
    text = OMM.synthetic_text(batch, 23 + batch)
    torch.manual_seed(24)
    masks = OMM.draw_decoder_masks(batch) if train else None
    o_loss, o_mu, o_lv, o_words, fed = oracle_objective(o_enc, o_dec, text, masks)
    o_loss.backward()
    loss, mu, logvar, words = hip_objective(enc, dec, text.to(DEV), masks)
    loss.backward()
    assert torch.equal(dec.last_fed.cpu(), fed), 'greedy feedback diverged'
    assert_close(mu, o_mu.detach(), 'mu'); assert_close(logvar, o_lv.detach(), 'logvar')
    assert_close(words, o_words.detach(), 'words'); assert_close(loss, o_loss.detach(), 'loss')
    worst = 0.0
    for (name, p), (_, q) in zip(list(enc.named_parameters()) + list(dec.named_parameters()),
                                 list(o_enc.named_parameters()) + list(o_dec.named_parameters())):
        worst = max(worst, assert_close(p.grad, q.grad, 'grad ' + name))
    print('text stacks B=%d D=%d %s: worst gradient rel err %.2e' % (batch, n_latents, 'train' if train else 'eval', worst))


def test_decoder_gradient_reaches_z_and_device_masks():
    """d words / d z (what the PoE backward receives) vs the oracle; without explicit masks the training-mode
    dropout draws come from the device stream: Bernoulli(0.9), fresh per call."""
    o_enc, o_dec, enc, dec = build(64, 31, 32, True)
    z = torch.randn(11, 64, generator=torch.Generator().manual_seed(33))
    torch.manual_seed(34)
    masks = OMM.draw_decoder_masks(11)
    zo = z.clone().requires_grad_()
    wo, _ = o_dec(zo, dropout_masks=masks)
    w8 = torch.randn(wo.shape, generator=torch.Generator().manual_seed(35))
    (wo * w8).sum().backward()
    zh = z.to(DEV).requires_grad_()
    wh = dec(zh, dropout_masks=masks)
    (wh * w8.to(DEV)).sum().backward()
    assert_close(wh, wo.detach(), 'words'); assert_close(zh.grad, zo.grad, 'd z')
    a = dec(z.to(DEV)); b = dec(z.to(DEV))
    assert not torch.equal(a, b), 'device dropout masks must differ between calls'
    m = torch.stack(dec._device_masks(4096))
    assert set(torch.unique(m).tolist()) <= {0.0, 1.0} and abs(m.mean().item() - 0.9) < 0.01


# ----------------------------------------------------------------------------- the K16 kernels one by one
@pytest.mark.parametrize('B,H', [(1, 200), (33, 200), (64, 8)])
def test_gru_cell_kernels(B, H):
    g = torch.Generator().manual_seed(B + H)
    gi, gh, hp = torch.randn(B, 3 * H, generator=g), torch.randn(B, 3 * H, generator=g), torch.randn(B, H, generator=g)
    gi_r, gh_r, hp_r = (t.clone().requires_grad_() for t in (gi, gh, hp))
    r = torch.sigmoid(gi_r[:, :H] + gh_r[:, :H]); z = torch.sigmoid(gi_r[:, H:2 * H] + gh_r[:, H:2 * H])
    n = torch.tanh(gi_r[:, 2 * H:] + r * gh_r[:, 2 * H:])
    h_ref = (1 - z) * n + z * hp_r
    dh = torch.randn(B, H, generator=g); dh2 = torch.randn(B, H, generator=g)
    h_ref.backward(dh + dh2)
    # strided views everywhere a leading dimension is accepted
    wide = torch.zeros(B, H + 7, device=DEV)
    hp_w = torch.zeros(B, H + 5, device=DEV); hp_w[:, :H] = hp.to(DEV)
    gates = torch.empty(B, 4 * H, device=DEV)
    K.gru_cell_fwd(gi.to(DEV), gh.to(DEV), hp_w[:, :H], wide[:, :H], gates)
    assert_close(wide[:, :H], h_ref.detach(), 'h_new', tol=1e-6)
    dgi, dgh, dhp = torch.empty(B, 3 * H, device=DEV), torch.empty(B, 3 * H, device=DEV), torch.empty(B, H, device=DEV)
    K.gru_cell_bwd(dh.to(DEV), dh2.to(DEV), gates, hp_w[:, :H], dgi, dgh, dhp)
    assert_close(dgi, gi_r.grad, 'dgi', tol=1e-5); assert_close(dgh, gh_r.grad, 'dgh', tol=1e-5)
    assert_close(dhp, hp_r.grad, 'dh_prev (direct part)', tol=1e-5)


def test_embedding_copy2d_argmax_kernels():
    g = torch.Generator().manual_seed(5)
    text = torch.randint(0, 12, (29, 4), generator=g)
    w = torch.randn(12, 200, generator=g)
    for swish in (False, True):
        wr = w.clone().requires_grad_()
        e = torch.nn.functional.embedding(text[:, 2], wr)
        if swish:
            e = e * torch.sigmoid(e)
        dout = torch.randn(29, 200, generator=g)
        e.backward(dout)
        out = torch.zeros(29, 264, device=DEV)
        K.embedding_fwd(text.to(DEV)[:, 2], w.to(DEV), out[:, :200], swish=swish)
        assert_close(out[:, :200], e.detach(), 'embedding fwd', tol=1e-6)
        assert float(out[:, 200:].abs().max()) == 0.0
        dwide = torch.zeros(29, 264, device=DEV); dwide[:, :200] = dout.to(DEV)
        dw = torch.empty(12, 200, device=DEV)
        K.embedding_bwd(text.to(DEV)[:, 2], w.to(DEV), dwide[:, :200], dw, swish=swish)
        assert_close(dw, wr.grad, 'embedding bwd', tol=1e-5)
        K.embedding_bwd(text.to(DEV)[:, 2], w.to(DEV), dwide[:, :200], dw, swish=swish, accumulate=True)
        assert_close(dw, 2 * wr.grad, 'embedding bwd accumulate', tol=1e-5)
    src = torch.randn(17, 40, generator=g); mask = (torch.rand(17, 24, generator=g) < 0.9).float()
    dst = torch.ones(17, 30, device=DEV)
    K.copy2d(src.to(DEV)[:, 8:32], dst[:, 3:27], mask=mask.to(DEV), scale=1 / 0.9, accumulate=True)
    ref = torch.ones(17, 30); ref[:, 3:27] += src[:, 8:32] * mask / 0.9
    assert_close(dst, ref, 'copy2d', tol=1e-6)
    x = torch.randn(50, 12, generator=g); x[3, 5] = x[3, 9] = 9.0       # a tie: the first maximum wins
    out = torch.empty(50, dtype=torch.int64, device=DEV)
    K.argmax_rows(x.to(DEV), out)
    assert torch.equal(out.cpu(), torch.max(torch.log_softmax(x, dim=1), dim=1)[1]) and out[3].item() == 5


def test_reference_names_and_errors():
    assert (MM.max_length, MM.n_characters, MM.SOS, MM.FILL) == (4, 12, 10, 11)
    assert MM.ProductOfExperts.VARIANT == 'B'
    x = torch.randn(5, 7)
    assert_close(MM.swish(x.to(DEV)), x * torch.sigmoid(x), 'swish', tol=1e-6)
    enc = MM.TextEncoder(8, MM.n_characters).to(DEV)
    with pytest.raises(ValueError):
        enc(torch.zeros(3, 4, device=DEV))
    with pytest.raises(RuntimeError, match='fused'):
        enc.gru(torch.zeros(4, 3, 200, device=DEV))
    dec = MM.TextDecoder(8, MM.n_characters).to(DEV).train()
    with pytest.raises(ValueError):
        dec(torch.zeros(3, 8, device=DEV), dropout_masks=[torch.ones(3, 200)] * 3)

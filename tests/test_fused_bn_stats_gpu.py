"""GPU, EXPERIMENTAL (skipped unless MVAE_EXPERIMENTAL=1): conv / transposed-conv forward launches that leave the
batch statistics of their output for the BatchNorm behind them (include/mvae_hip.h: mvae_conv*_k4_fwd_stats,
mvae_bn_train_fwd_parts; layers.forward_tape under MVAE_FUSED_BN_STATS=1).  The path was written and compiled in
round 2 without hardware left to run it; these tests are its acceptance bar (same as the path they would replace:
conv output bit-equal, BatchNorm output / saved statistics 1e-5, the fused steps against the live oracle at the
north_star tolerance, running statistics 1e-5).  Run first thing:
    MVAE_EXPERIMENTAL=1 python -m pytest tests/test_fused_bn_stats_gpu.py -m gpu -q"""
import os

import numpy as np
import pytest
import torch

import mvae_amd
from mvae_amd import kernels as K
from mvae_amd.engine import BimodalStep, Celeba19Step, sample_subsets
from oracle import steps as OS
from test_engine_gpu import build_pair, check_bn_vs, check_grads_vs_oracle
from util import assert_close

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('MVAE_EXPERIMENTAL') != '1',
                                 reason='experimental path, not yet verified on hardware (MVAE_EXPERIMENTAL=1 runs it)')]
DEV = 'cuda'


def g(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# transposed, G, B, Cin, H, Cout, stride, pad   -- every tile shape the conv forms use, one group and several
STATS_CASES = [(0, 1, 8, 32, 32, 64, 2, 1), (0, 1, 8, 64, 16, 128, 2, 1), (0, 1, 64, 128, 8, 256, 1, 0),
               (1, 3, 8, 128, 4, 64, 2, 1), (1, 3, 8, 64, 8, 32, 2, 1), (1, 1, 16, 64, 8, 32, 2, 1),
               (0, 1, 256, 32, 32, 64, 2, 1), (1, 3, 256, 64, 8, 32, 2, 1), (1, 19, 8, 64, 8, 32, 2, 1),
               (0, 2, 16, 32, 16, 32, 2, 1), (1, 2, 64, 256, 4, 128, 2, 1)]


@pytest.mark.parametrize('tr,G,B,Cin,H,Cout,s,p', STATS_CASES)
def test_conv_stats_then_bn_equals_conv_then_bn(tr, G, B, Cin, H, Cout, s, p):
    x = (g(G * B, Cin, H, H, seed=40) + 0.3).to(DEV)
    w = (g(Cin, Cout, 4, 4, seed=41, scale=(Cin * 4) ** -0.5) if tr else
         g(Cout, Cin, 4, 4, seed=41, scale=(Cin * 16) ** -0.5)).to(DEV)
    OH = (H - 1) * s - 2 * p + 4 if tr else (H + 2 * p - 4) // s + 1
    shape = (G * B, Cout, OH, OH)
    lay = K.conv_stats_layout(tr, G * B, Cin, H, H, Cout, s, p)
    assert lay is not None and lay.tiles_j % G == 0, 'pick a case that has a statistics launch'
    gamma, beta = (1 + 0.1 * g(Cout, seed=42)).to(DEV), (0.1 * g(Cout, seed=43)).to(DEV)

    def bn_state():
        return (torch.empty(shape, device=DEV), torch.empty(G, Cout, device=DEV), torch.empty(G, Cout, device=DEV),
                (0.05 * g(Cout, seed=44)).to(DEV), (1 + 0.1 * g(Cout, seed=45).abs()).to(DEV))
    # the path in use: forward, then BatchNorm sweeps the tensor for its statistics
    pre_a = torch.empty(shape, device=DEV)
    (K.convT2d_fwd if tr else K.conv2d_fwd)(x, w, pre_a, None, s, p)
    ya, sma, sia, rma, rva = bn_state()
    K.bn_train_fwd(pre_a, gamma, beta, ya, sma, sia, rma, rva, G, n_updates=2)
    # statistics from the conv epilogue
    pre_b = torch.empty(shape, device=DEV)
    rec = torch.full((lay.parts() * 2 * Cout,), float('nan'), device=DEV)
    lay_b = (K.convT2d_fwd_stats if tr else K.conv2d_fwd_stats)(x, w, pre_b, s, p, rec)
    assert (lay_b.ncls, lay_b.tiles_j, lay_b.ppt, lay_b.cols) == (lay.ncls, lay.tiles_j, lay.ppt, lay.cols)
    assert torch.equal(pre_a, pre_b), 'the statistics launch must store the same conv output'
    assert torch.isfinite(rec).all(), 'every record written'
    yb, smb, sib, rmb, rvb = bn_state()
    K.bn_train_fwd_parts(pre_b, gamma, beta, yb, smb, sib, rmb, rvb, G, shape, rec, lay_b, n_updates=2)
    for a, b, what in ((smb, sma, 'saved mean'), (sib, sia, 'saved invstd'), (rmb, rma, 'running mean'),
                       (rvb, rva, 'running var'), (yb, ya, 'output')):
        assert_close(a, b, what, tol=1e-5)
    # and against float64 statistics of the stored tensor
    t = pre_a.double().reshape(G, B, Cout, -1)
    mean = t.mean(dim=(1, 3)); var = t.var(dim=(1, 3), unbiased=False)
    assert_close(smb, mean, 'saved mean vs float64', tol=1e-5)
    assert_close(sib, (var + 1e-5).rsqrt(), 'saved invstd vs float64', tol=1e-5)
    # statistics only: nothing stored, same running statistics (deterministic: bit-equal to the stored run)
    rec2 = torch.empty_like(rec)
    lay_c = (K.convT2d_fwd_stats if tr else K.conv2d_fwd_stats)(x, w, None, s, p, rec2)
    assert torch.equal(rec, rec2)
    _, smc, sic, rmc, rvc = bn_state()
    K.bn_train_fwd_parts(None, gamma, beta, None, smc, sic, rmc, rvc, G, shape, rec2, lay_c, n_updates=2)
    assert torch.equal(rmc, rmb) and torch.equal(rvc, rvb) and torch.equal(smc, smb)


def test_stats_launch_refuses_shapes_without_one():
    x = g(7, 128, 8, 8, seed=50).to(DEV); w = g(256, 128, 4, 4, seed=51).to(DEV)        # 7 * 25 columns: ragged tile
    assert K.conv_stats_layout(0, 7, 128, 8, 8, 256, 1, 0) is None
    with pytest.raises(RuntimeError, match='MVAE_ERR_ARG'):
        K.conv2d_fwd_stats(x, w, torch.empty(7, 256, 5, 5, device=DEV), 1, 0, torch.empty(1 << 16, device=DEV))
    x = g(8, 32, 32, 32, seed=52).to(DEV); w = g(64, 32, 4, 4, seed=53).to(DEV)
    with pytest.raises(RuntimeError, match='MVAE_ERR_WS'):
        K.conv2d_fwd_stats(x, w, torch.empty(8, 64, 16, 16, device=DEV), 2, 1, torch.empty(16, device=DEV))


@pytest.mark.parametrize('batch', [8, 64])
def test_celeba_step_with_fused_statistics_matches_live_oracle(batch, monkeypatch):
    monkeypatch.setenv('MVAE_FUSED_BN_STATS', '1')
    kind = 'celeba'
    oracle, model, d = build_pair(kind, weight_seed=11)
    image, label = OS.synthetic_batch(kind, batch, seed=77)
    torch.manual_seed(5)
    noise = OS.draw_bimodal_noise(batch, d, has_dropout=True)
    total, terms, lat = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, 10.0, 0.37)
    total.backward()
    eng = BimodalStep(model, batch, 1.0, 10.0)
    elbo = eng.terms_in_reference_order(eng.step(image.to(DEV), label.to(DEV), 0.37, noise=noise)).cpu()
    used = [m for m in model.modules() if getattr(m, '_stats_lay', None) and any(v is not None for v in m._stats_lay.values())]
    assert len(used) >= 4, 'the fused statistics launches were not taken'
    assert_close(elbo[:3], torch.stack(terms).detach(), 'ELBO terms')
    assert_close(elbo[3], total.detach(), 'total')
    check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())


def test_celeba19_step_with_fused_statistics_matches_live_oracle(monkeypatch):
    monkeypatch.setenv('MVAE_FUSED_BN_STATS', '1')
    batch, approx_m = 8, 1
    oracle, model, d = build_pair('celeba19', weight_seed=23)
    image, attrs = OS.synthetic_batch('celeba19', batch, seed=81)
    combos = sample_subsets(np.random.RandomState(6), 19, approx_m)
    combos[:, 0] = True
    if combos[0].sum() < 2:
        combos[0, 1:3] = True
    terms = OS.celeba19_terms(combos)
    torch.manual_seed(9)
    noise = OS.draw_celeba19_noise(batch, d, terms)
    total, elbos, lat = OS.celeba19_step(oracle, image, attrs, terms, noise, 1.0, 10.0, 0.3)
    total.backward()
    eng = Celeba19Step(model, batch, 1.0, 10.0, approx_m=approx_m)
    elbo = eng.step(image.to(DEV), attrs.to(DEV), 0.3, noise=noise, combos=combos).cpu()
    T = len(terms)
    assert_close(elbo[:T], torch.stack(elbos).detach(), 'ELBO terms')
    assert_close(elbo[T], total.detach(), 'total')
    check_grads_vs_oracle(model, oracle)
    check_bn_vs(model, oracle.state_dict())      # includes the 18 statistics-only decodes (nothing stored there)


def test_captured_celeba_step_is_the_same_with_and_without_fused_statistics(monkeypatch):
    """Graph replay, three steps, MVAE_FUSED_BN_STATS=1 vs 0: the parameters agree to round-off of the statistics'
    summation order (1e-5 of the total movement) and the fused run is reproducible bit for bit."""
    from mvae_amd.optim import FusedAdam
    finals = []
    for flag in ('1', '1', '0'):
        monkeypatch.setenv('MVAE_FUSED_BN_STATS', flag)
        _, model, _ = build_pair('celeba', weight_seed=29)
        opt = FusedAdam(model.parameters(), lr=1e-4)
        eng = BimodalStep(model, 16, 1.0, 10.0)
        image, label = OS.synthetic_batch('celeba', 16, seed=700)
        eng.capture(opt, image.shape[1:], label)
        for step in range(3):
            image, label = OS.synthetic_batch('celeba', 16, seed=710 + step)
            eng.replay(image.to(DEV), label.to(DEV), 0.5)
        torch.cuda.synchronize()
        finals.append(model.arena.flat.clone())
    assert torch.equal(finals[0], finals[1])
    assert (finals[0] - finals[2]).abs().max().item() <= 0.05 * 3 * 1e-4

"""CPU: size-independent properties of the oracle's pieces (what the reference's formulas imply),
complementing the golden fixtures: product of experts, KL, the hand-written losses, the ELBO's
linearity in its coefficients (mnist/model.py:156-185, mnist/train.py:20-94)."""
import pytest
import torch

from oracle import functional as OF
from oracle import models as OM
from oracle import steps as OS


def test_poe_single_expert_with_prior_closed_form():
    """prior N(0,1) x expert N(m, v): precision-weighted mean m/(1+v), variance v/(1+v) (up to the
    1e-8 guards of variant B, celeba/model.py:200-207)."""
    g = torch.Generator().manual_seed(0)
    m = torch.randn(5, 7, generator=g); lv = torch.randn(5, 7, generator=g) * 0.5
    mu, logvar = OF.poe_with_prior([m], [lv], 'B')
    v = lv.exp()
    assert torch.allclose(mu, m / (1 + v), rtol=1e-5, atol=1e-6)
    assert torch.allclose(logvar.exp(), v / (1 + v), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('variant', ['A', 'B'])
def test_poe_is_permutation_invariant_and_sharpens(variant):
    g = torch.Generator().manual_seed(1)
    mus = [torch.randn(4, 6, generator=g) for _ in range(3)]
    lvs = [torch.randn(4, 6, generator=g) * 0.3 for _ in range(3)]
    a = OF.poe_with_prior(mus, lvs, variant)
    b = OF.poe_with_prior(mus[::-1], lvs[::-1], variant)
    assert torch.allclose(a[0], b[0], atol=1e-6) and torch.allclose(a[1], b[1], atol=1e-6)
    # the product is never wider than its narrowest factor (the unit prior included)
    assert (a[1] <= torch.stack(lvs + [torch.zeros(4, 6)]).min(0).values + 1e-4).all()


def test_kl_rows_zero_at_the_prior_and_positive_elsewhere():
    assert torch.equal(OF.kl_rows(torch.zeros(3, 8), torch.zeros(3, 8)), torch.zeros(3))
    g = torch.Generator().manual_seed(2)
    kl = OF.kl_rows(torch.randn(16, 8, generator=g), torch.randn(16, 8, generator=g))
    assert (kl > 0).all()


def test_bce_with_logits_matches_the_definition_and_its_symmetry():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(9, 11, generator=g) * 4; t = torch.rand(9, 11, generator=g)
    got = OF.binary_cross_entropy_with_logits(x, t)
    p = torch.sigmoid(x.double())
    ref = -(t.double() * p.log() + (1 - t.double()) * (1 - p).log())
    assert torch.allclose(got.double(), ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(OF.binary_cross_entropy_with_logits(-x, 1 - t), got, atol=1e-6)   # bce(x,t) = bce(-x,1-t)


def test_cross_entropy_is_shift_invariant_and_picks_the_label():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(6, 10, generator=g); y = torch.randint(0, 10, (6,), generator=g)
    a = OF.cross_entropy(x, y).sum(1)
    b = OF.cross_entropy(x + 3.0, y).sum(1)
    assert torch.allclose(a, b, atol=1e-5)
    ref = -torch.log_softmax(x + 1e-6, dim=1)[torch.arange(6), y]
    assert torch.allclose(a, ref, atol=1e-6)


def test_elbo_is_linear_in_its_coefficients():
    """total(lambda_image, lambda_label, beta) = li*A + ll*B + beta*C for fixed noise: three evaluations
    determine A, B, C; a fourth must agree (mnist/train.py:57)."""
    cls, d = OM.MODELS['mnist']
    model = OM.fill_parameters(cls(d), 5).train()
    image, label = OS.synthetic_batch('mnist', 6, seed=7)
    torch.manual_seed(8)
    noise = OS.draw_bimodal_noise(6, d, has_dropout=False)

    def total(li, ll, beta):
        with torch.no_grad():
            return OS.bimodal_step(model, 'mnist', image, label, noise, li, ll, beta)[0].item()
    A, B, C = total(1, 0, 0), total(0, 1, 0), total(0, 0, 1)
    assert abs(total(2.0, 50.0, 0.3) - (2.0 * A + 50.0 * B + 0.3 * C)) < 1e-3 * abs(50.0 * B)

"""The reference's module-level names on the drop-in ``model.py`` files (missing item 3):
``ProductOfExperts`` (callable on a stacked [M, B, D] pair), ``prior_expert``, ``Swish`` and the ``model.experts``
attribute -- mnist/model.py:26,149-185, celeba/model.py:25,193-229 -- against the oracle, forward and backward."""
import pytest
import torch

import mvae_amd
from oracle import functional as OF
from oracle import models as OM
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'
VARIANT = {'mnist': 'A', 'fashionmnist': 'A', 'celeba': 'B', 'celeba19': 'B'}


@pytest.mark.parametrize('kind', sorted(VARIANT))
@pytest.mark.parametrize('M,B,D', [(2, 5, 64), (3, 33, 100), (20, 7, 100)])
def test_product_of_experts_on_a_stacked_pair(kind, M, B, D):
    mod = getattr(mvae_amd, kind).model
    poe = mod.ProductOfExperts()
    assert poe.VARIANT == VARIANT[kind]
    g = torch.Generator().manual_seed(M * 100 + B)
    pm, plv = mod.prior_expert((1, B, D), use_cuda=True)
    assert pm.is_cuda and pm.shape == (1, B, D) and float(pm.abs().max()) == 0 and float(plv.abs().max()) == 0
    mu_e = torch.randn(M - 1, B, D, generator=g)
    lv_e = 0.5 * torch.randn(M - 1, B, D, generator=g)
    # the reference's infer(): torch.cat the experts behind the prior, then self.experts(mu, logvar)
    mu_ref = torch.cat([torch.zeros(1, B, D), mu_e]).requires_grad_()
    lv_ref = torch.cat([torch.zeros(1, B, D), lv_e]).requires_grad_()
    r_mu, r_lv = OF.poe(mu_ref, lv_ref, VARIANT[kind])
    w1, w2 = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    ((r_mu * w1).sum() + (r_lv * w2).sum()).backward()
    mu = torch.cat([pm, mu_e.to(DEV)]).requires_grad_()
    lv = torch.cat([plv, lv_e.to(DEV)]).requires_grad_()
    h_mu, h_lv = poe(mu, lv)
    assert h_mu.shape == (B, D)
    assert_close(h_mu, r_mu.detach(), 'pd_mu', tol=1e-5)
    assert_close(h_lv, r_lv.detach(), 'pd_logvar', tol=1e-5)
    ((h_mu * w1.to(DEV)).sum() + (h_lv * w2.to(DEV)).sum()).backward()
    assert_close(mu.grad, mu_ref.grad, 'd mu stack', tol=1e-4)
    assert_close(lv.grad, lv_ref.grad, 'd logvar stack', tol=1e-4)
    # [M, D] input, as the reference's docstring describes it
    f_mu, f_lv = poe(mu.detach()[:, 0], lv.detach()[:, 0])
    assert f_mu.shape == (D,)
    assert_close(f_mu, r_mu.detach()[0], 'pd_mu 2-D', tol=1e-5)
    assert_close(f_lv, r_lv.detach()[0], 'pd_logvar 2-D', tol=1e-5)


@pytest.mark.parametrize('kind', ['mnist', 'celeba'])
def test_model_experts_reproduces_infer(kind):
    """``model.experts`` on the stack the reference's infer() builds == the drop-in's fused ``model.infer``."""
    cls, d = OM.MODELS[kind]
    oracle = OM.fill_parameters(cls(d), 3).eval()
    mod = getattr(mvae_amd, kind).model
    model = mod.MVAE(d)
    model.load_state_dict(oracle.state_dict())
    model.to(DEV).eval()
    assert isinstance(model.experts, mod.ProductOfExperts)
    from oracle import steps as OS
    image, label = OS.synthetic_batch(kind, 6, seed=4)
    image, label = image.to(DEV), label.to(DEV)
    with torch.no_grad():
        mu, logvar = mod.prior_expert((1, 6, d), use_cuda=True)
        i_mu, i_lv = model.image_encoder(image)
        l_mu, l_lv = (model.text_encoder if kind == 'mnist' else model.attrs_encoder)(label)
        mu = torch.cat((mu, i_mu.unsqueeze(0), l_mu.unsqueeze(0)), dim=0)
        logvar = torch.cat((logvar, i_lv.unsqueeze(0), l_lv.unsqueeze(0)), dim=0)
        a_mu, a_lv = model.experts(mu, logvar)
        b_mu, b_lv = model.infer(image, label)
    assert_close(a_mu, b_mu, 'mu', tol=1e-6)
    assert_close(a_lv, b_lv, 'logvar', tol=1e-6)


def test_swish_module_matches_oracle():
    from mvae_amd.mnist.model import Swish
    x = torch.randn(37, 19)
    assert_close(Swish()(x.to(DEV)), OF.swish(x), 'swish', tol=1e-6)


def test_bad_arguments():
    poe = mvae_amd.mnist.model.ProductOfExperts()
    z = torch.zeros(2, 3, 4, device=DEV)
    with pytest.raises(ValueError):
        poe(z, z, eps=1e-6)
    with pytest.raises(ValueError):
        poe(z, z[:, :2])
    with pytest.raises(RuntimeError, match='GPU'):
        poe(z.cpu(), z.cpu())

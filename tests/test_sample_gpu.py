"""GPU: conditional generation (mnist/sample.py:71-113, celeba/sample.py) on the drop-in modules in
eval mode vs the oracle: posterior of each conditioning pattern, decoded samples, and the CLI."""
import os
import subprocess
import sys

import pytest
import torch

import mvae_amd  # noqa: F401
from mvae_amd.sample_common import generate, posterior
from oracle import steps as OS
from test_engine_gpu import build_pair
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('kind', ['mnist', 'fashionmnist', 'celeba'])
def test_generation_matches_oracle(kind):
    oracle, model, d = build_pair(kind, weight_seed=61)
    if kind == 'celeba':        # give the BatchNorm running statistics some non-trivial content
        g = torch.Generator().manual_seed(5)
        for (n, b), (_, bo) in zip(model.named_buffers(), oracle.named_buffers()):
            if n.endswith('running_mean'):
                v = torch.randn(b.shape, generator=g) * 0.1
            elif n.endswith('running_var'):
                v = torch.rand(b.shape, generator=g) + 0.5
            else:
                continue
            b.copy_(v.to(DEV)); bo.copy_(v)
    oracle.eval(); model.eval()
    image, label = OS.synthetic_batch(kind, 1, seed=3)
    eps = torch.randn(16, d, generator=torch.Generator().manual_seed(9))
    for use_img, use_lbl in [(False, False), (True, False), (False, True), (True, True)]:
        img = image.to(DEV) if use_img else None
        lbl = label.to(DEV) if use_lbl else None
        mu, std = posterior(model, img, lbl)
        z, img_p, lbl_logits = generate(model, 16, mu, std, eps=eps)
        with torch.no_grad():
            if use_img or use_lbl:
                mu_o, lv_o = oracle.infer(image=image if use_img else None, label=label if use_lbl else None)
                std_o = lv_o.mul(0.5).exp()
            else:
                mu_o, std_o = torch.zeros(1), torch.ones(1)
            z_o = eps * std_o.expand(16, d) + mu_o.expand(16, d)
            img_o = torch.sigmoid(oracle.image_decoder(z_o))
            lbl_o = getattr(oracle, 'attrs_decoder' if kind == 'celeba' else 'text_decoder')(z_o)
        tag = '%s img=%d lbl=%d' % (kind, use_img, use_lbl)
        assert_close(z, z_o, tag + ' z')
        assert_close(img_p.reshape(16, -1), img_o.reshape(16, -1), tag + ' image probabilities')
        assert_close(lbl_logits, lbl_o, tag + ' label logits')


@pytest.mark.parametrize('kind,flags', [('mnist', ['--condition-on-text', '3']),
                                         ('fashionmnist', ['--condition-on-image', '2', '--synthetic']),
                                         ('celeba', ['--condition-on-text', 'Smiling'])])
def test_sample_cli(kind, flags, tmp_path):
    pkg = os.path.join(ROOT, 'multimodal-vae-public_amd', kind)
    lam = ['--lambda-attrs', '10'] if kind == 'celeba' else ['--lambda-text', '50']
    out = subprocess.run([sys.executable, os.path.join(pkg, 'train.py'), '--cuda', '--synthetic', '--epochs', '1',
                          '--steps-per-epoch', '3', '--batch-size', '8', '--n-latents', '16', '--out-dir',
                          str(tmp_path)] + lam, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    out = subprocess.run([sys.executable, os.path.join(pkg, 'sample.py'), os.path.join(str(tmp_path), 'model_best.pth.tar'),
                          '--cuda', '--n-samples', '16', '--out-dir', str(tmp_path)] + flags,
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert os.path.getsize(os.path.join(str(tmp_path), 'sample_image.png')) > 500
    txt = 'sample_attrs.txt' if kind == 'celeba' else 'sample_text.txt'
    assert len(open(os.path.join(str(tmp_path), txt)).read().splitlines()) == 16

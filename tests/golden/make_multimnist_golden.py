#!/usr/bin/env python
"""Golden fixture of the MultiMNIST text stacks from the UNMODIFIED reference
(multimnist/model.py:145-235 TextEncoder / TextDecoder, multimnist/train.py:47-58,100-117 text loss).

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_multimnist_golden.py

Imports the reference's ``multimnist/model.py`` and ``multimnist/train.py`` (import-time shims only: the
py2 name ``xrange``, stub ``torchvision`` / ``tqdm`` / ``datasets`` modules that train.py imports but the
text path never touches), fills both stacks with the deterministic weights of
``oracle.models.fill_parameters``, runs in TRAINING mode under ``torch.manual_seed``

    mu, logvar = TextEncoder(text);  z = mu + 0.5 * logvar;  words = TextDecoder(z)
    loss = mean_b( sum_digits sum_classes cross_entropy(words, text) ) + 0.1 * mean(mu^2 + logvar^2)

(the reference's own ``cross_entropy``), backward, and records the inputs, the four dropout masks the global
generator produced inside nn.GRU (re-drawn in order), outputs, the fed-back characters, the loss and every
parameter's gradient digest.  It asserts that the oracle restatement (``oracle/multimnist.py``) reproduces
every recorded value before anything is written.  The fixture is data (npz); no reference source text is stored."""
import builtins
import importlib
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/multimnist'

from oracle import models as OM, multimnist as OMM  # noqa: E402

N_LATENTS, BATCH = 64, 6
ENC_SEED, DEC_SEED, TEXT_SEED, NOISE_SEED = 11, 12, 13, 14


def import_reference():
    builtins.xrange = range
    for name, attrs in (('torchvision', {'transforms': types.ModuleType('torchvision.transforms')}),
                        ('tqdm', {'tqdm': lambda x, **k: x}), ('datasets', {'MultiMNIST': object})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    sys.modules['torchvision.transforms'] = sys.modules['torchvision'].transforms
    for name in ('model', 'train', 'utils'):
        sys.modules.pop(name, None)
    sys.path.insert(0, REF)
    try:
        model = importlib.import_module('model')
        train = importlib.import_module('train')
        utils = importlib.import_module('utils')
    finally:
        sys.path.remove(REF)
    return model, train, utils


def grad_digest(prefix, module, out):
    for name, p in module.named_parameters():
        g = p.grad.detach().reshape(-1)
        out['gnorm/%s.%s' % (prefix, name)] = np.float64(g.double().norm().item())
        out['ghead/%s.%s' % (prefix, name)] = g[:8].numpy().copy()


def objective(enc, dec, text, ce, dec_kwargs):
    mu, logvar = enc(text)
    z = mu + 0.5 * logvar
    words = dec(z, **dec_kwargs)
    fed = None
    if isinstance(words, tuple):
        words, fed = words
    B, L, K = words.shape
    rows = ce(words.reshape(-1, K), text.reshape(-1)).sum(dim=1).view(B, L).sum(dim=1)
    loss = rows.mean() + 0.1 * (mu.pow(2) + logvar.pow(2)).mean()
    return loss, mu, logvar, words, fed


def main():
    warnings.simplefilter('ignore')
    M, T, U = import_reference()
    assert (U.max_length, U.n_characters, U.SOS, U.FILL) == (OMM.MAX_LENGTH, OMM.N_CHARACTERS, OMM.SOS, OMM.FILL)
    ref_enc = OM.fill_parameters(M.TextEncoder(N_LATENTS, U.n_characters, n_hiddens=200, bidirectional=True), ENC_SEED).train()
    ref_dec = OM.fill_parameters(M.TextDecoder(N_LATENTS, U.n_characters, n_hiddens=200), DEC_SEED).train()
    text = OMM.synthetic_text(BATCH, TEXT_SEED)

    torch.manual_seed(NOISE_SEED)
    loss, mu, logvar, words, _ = objective(ref_enc, ref_dec, text, T.cross_entropy, {})
    loss.backward()

    # the oracle on the masks the generator produced, re-drawn in the reference's order
    enc = OMM.TextEncoder(N_LATENTS); enc.load_state_dict(ref_enc.state_dict()); enc.train()
    dec = OMM.TextDecoder(N_LATENTS); dec.load_state_dict(ref_dec.state_dict()); dec.train()
    torch.manual_seed(NOISE_SEED)
    masks = OMM.draw_decoder_masks(BATCH)
    from oracle.functional import cross_entropy
    o_loss, o_mu, o_lv, o_words, fed = objective(enc, dec, text, cross_entropy, {'dropout_masks': masks})
    o_loss.backward()

    def close(a, b, what, tol=2e-6):
        err = (a.detach() - b.detach()).abs().max().item() / max(b.detach().abs().max().item(), 1e-30)
        assert err <= tol, '%s: oracle vs reference %.3e' % (what, err)
    close(o_mu, mu, 'mu'); close(o_lv, logvar, 'logvar'); close(o_words, words, 'words'); close(o_loss, loss, 'loss')
    assert torch.equal(OMM.text_loss_rows(o_words, text).mean() + 0.1 * (o_mu.pow(2) + o_lv.pow(2)).mean(), o_loss)
    for (n, p), (_, q) in zip(list(ref_enc.named_parameters()) + list(ref_dec.named_parameters()),
                              list(enc.named_parameters()) + list(dec.named_parameters())):
        close(q.grad, p.grad, 'grad ' + n, tol=2e-5)
    # eval mode: no dropout, same feedback rule
    ref_dec.eval(); dec.eval()
    with torch.no_grad():
        zz = mu.detach() + 0.5 * logvar.detach()
        close(dec(zz)[0], ref_dec(zz), 'eval words')

    fx = {'text': text.numpy(), 'mu': mu.detach().numpy(), 'logvar': logvar.detach().numpy(),
          'words': words.detach().numpy(), 'fed': fed.numpy(), 'loss': np.float64(loss.item()),
          'eval_words': ref_dec(zz).detach().numpy()}
    for i, m in enumerate(masks):
        fx['mask%d' % i] = m.numpy().astype(np.uint8)
    grad_digest('text_encoder', ref_enc, fx)
    grad_digest('text_decoder', ref_dec, fx)
    fx['meta'] = np.array(repr({'n_latents': N_LATENTS, 'batch': BATCH, 'enc_seed': ENC_SEED, 'dec_seed': DEC_SEED,
                                'text_seed': TEXT_SEED, 'noise_seed': NOISE_SEED, 'torch': torch.__version__}))
    path = os.path.join(HERE, 'multimnist_text.npz')
    np.savez_compressed(path, **fx)
    print('wrote %s (%d arrays, %d bytes); loss %.6f' % (path, len(fx), os.path.getsize(path), loss.item()))


if __name__ == '__main__':
    main()

"""GPU: the reference's ``test(epoch)`` body through the drop-in modules, against values captured from the real
reference (tests/golden/make_eval_golden.py -> eval_<experiment>.npz; the oracle is pinned to the same files on CPU
in test_oracle_golden.py).  Written after the round's GPU minutes were spent -- the path it checks is the one
test_engine_gpu.py::test_eval_mode_and_error_behaviour and the train.py CLI tests already run, the reference
values are new -- so the file is named to run last under ``pytest -x``."""
import numpy as np
import pytest
import torch

import mvae_amd
from oracle import models as OM, steps as OS
from util import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('kind', ['mnist', 'fashionmnist', 'celeba', 'celeba19'])
def test_eval_pass_matches_reference_golden(golden_dir, kind):
    """The reference's test(epoch) body -- eval mode, BatchNorm on running statistics, reparametrize = mu -- through
    the drop-in module surface and the product's own ``_test_total`` (what ``train.py`` prints as Test Loss),
    against values captured from the real reference (tests/golden/make_eval_golden.py)."""
    import argparse
    import importlib
    fx, meta = load_golden(golden_dir, 'eval_' + kind)
    cls, d = OM.MODELS[kind]
    sd = OM.fill_parameters(cls(d), meta['weight_seed']).state_dict()
    for k in list(sd):
        if 'running_' in k or 'num_batches' in k:
            sd[k] = torch.from_numpy(np.asarray(fx['bn/' + k])).to(sd[k].dtype)
    model = getattr(mvae_amd, kind).model.MVAE(d)
    model.load_state_dict(sd)
    model.to(DEV).train()
    model.finalize()
    model.eval()
    image, label = OS.synthetic_batch(kind, meta['batch'], meta['input_seed'])
    img, lbl = image.to(DEV), label.to(DEV)
    args = argparse.Namespace(lambda_image=meta['lambda_image'], lambda_attrs=meta['lambda_label'],
                              lambda_text=meta['lambda_label'])
    test_total = importlib.import_module('mvae_amd.%s.train' % kind)._test_total
    with torch.no_grad():
        total = test_total(model, img, lbl, args)
        if kind == 'celeba19':
            calls = [model(img, [lbl[:, i] for i in range(lbl.shape[1])])]
            calls = [(calls[0][0], torch.stack(calls[0][1], dim=1), calls[0][2], calls[0][3])]
        elif kind == 'celeba':
            calls = [model(img, lbl), model(img), model(attrs=lbl)]
        else:
            calls = [model(img, lbl), model(img), model(text=lbl)]
    assert_close(total.item(), fx['total'], 'test loss')
    for c, r in enumerate(calls):
        assert_close(r[2], fx['mu%d' % c], 'eval mu%d' % c)
        assert_close(r[3], fx['logvar%d' % c], 'eval logvar%d' % c)
        assert_close(r[0][0].reshape(-1)[:256], fx['logits_image_0_%d' % c], 'eval image logits %d' % c)
        assert_close(r[1], fx['logits_label_%d' % c], 'eval label logits %d' % c)

#!/usr/bin/env python
"""Generate eval_<experiment>.npz from the UNMODIFIED reference: the body of its ``test(epoch)`` closure.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_eval_golden.py

The reference's test loop (mnist/train.py:229-253, fashionmnist/train.py and celeba/train.py the same text;
celeba19/train.py:319-340) puts the model in eval mode -- reparametrize returns mu, Dropout is off, BatchNorm
uses its running statistics -- calls it like the train step does (three calls; celeba19: the joint call only) and
sums the reference's own ``elbo_loss`` at annealing 1 and its DEFAULT lambdas -- except celeba/train.py:238-243,
which passes the CLI's (--lambda-image 1, --lambda-attrs 10 by default; used here).  This script runs that
body on a batch of 6 with the deterministic weights of ``oracle.models.fill_parameters``, after one train-mode
joint forward that moves the BatchNorm running statistics off their initial 0 / 1, and records the inputs, the
BatchNorm state the eval pass read, every call's mu / logvar, logits of sample 0 and the losses.  It asserts the
oracle restatement reproduces each value before writing.  The fixtures are data; no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import LAMBDA_LABEL, WEIGHT_SEED, bn_stats, check, import_reference, save   # noqa: E402
from oracle import functional as OF, models as OM, steps as OS  # noqa: E402

BATCH = 6


def move_running_stats(ref, exp, image, label):
    """One train-mode joint forward (noise from a fixed seed): BatchNorm running statistics become data-dependent."""
    if not any('running_' in k for k in ref.state_dict()):
        return
    ref.train()
    torch.manual_seed(77)
    with torch.no_grad():
        if exp == 'celeba19':
            ref(image, [label[:, i] for i in range(label.shape[1])])
        else:
            ref(image, label)


def run(exp):
    ref_model_mod, ref_train = import_reference(exp)
    cls, d = OM.MODELS[exp]
    oracle = OM.fill_parameters(cls(d), WEIGHT_SEED)
    ref = ref_model_mod.MVAE(d)
    ref.load_state_dict(oracle.state_dict())
    image, label = OS.synthetic_batch(exp, BATCH, seed=4321)
    move_running_stats(ref, exp, image, label)
    oracle.load_state_dict(ref.state_dict())
    ref.eval(); oracle.eval()
    fx = {'label': label.numpy()}
    fx.update(bn_stats(ref))
    with torch.no_grad():
        if exp == 'celeba19':
            attrs = ref_train.tensor_2d_to_list(label)
            ri, ra, mu, lv = ref(image, attrs)
            total = ref_train.elbo_loss([ri] + ra, [image] + attrs, mu, lv)
            calls = [(ri, torch.stack(ra, dim=1), mu, lv)]
            terms = [total]
            o_attrs = [label[:, i] for i in range(label.shape[1])]
            ori, ora, omu, olv, oz = oracle(image, o_attrs)
            o_total = OF.elbo_loss_multi([ori] + ora, [image] + o_attrs, omu, olv)
            o_calls = [(ori, torch.stack(ora, dim=1), omu, olv, oz)]
            o_terms = [o_total]
        else:
            kw = 'attrs' if exp == 'celeba' else 'text'
            lam = dict(lambda_image=1.0, lambda_attrs=LAMBDA_LABEL[exp]) if exp == 'celeba' else {}
            r1, r2, r3 = ref(image, label), ref(image), ref(**{kw: label})
            joint = ref_train.elbo_loss(r1[0], image, r1[1], label, r1[2], r1[3], **lam)
            img = ref_train.elbo_loss(r2[0], image, None, None, r2[2], r2[3], **lam)
            lbl = ref_train.elbo_loss(None, None, r3[1], label, r3[2], r3[3], **lam)
            total = joint + img + lbl
            calls, terms = [r1, r2, r3], [joint, img, lbl]
            elbo = OF.elbo_loss_attrs if exp == 'celeba' else OF.elbo_loss_label
            o1, o2, o3 = oracle(image, label), oracle(image, None), oracle(None, label)
            o_terms = [elbo(o1[0], image, o1[1], label, o1[2], o1[3], **lam),
                       elbo(o2[0], image, None, None, o2[2], o2[3], **lam),
                       elbo(None, None, o3[1], label, o3[2], o3[3], **lam)]
            o_total = o_terms[0] + o_terms[1] + o_terms[2]
            o_calls = [o1, o2, o3]
    fx['total'] = np.float64(total.item())
    fx['terms'] = np.array([t.item() for t in terms], dtype=np.float64)
    for c, (r, o) in enumerate(zip(calls, o_calls)):
        fx['mu%d' % c] = r[2].numpy().copy()
        fx['logvar%d' % c] = r[3].numpy().copy()
        fx['logits_image_0_%d' % c] = r[0][0].numpy().reshape(-1)[:256].copy()
        fx['logits_label_%d' % c] = r[1].numpy().copy()
        check(o[2], r[2], '%s eval mu%d' % (exp, c))
        check(o[3], r[3], '%s eval logvar%d' % (exp, c))
        check(o[4], r[2], '%s eval z%d = mu' % (exp, c))                 # eval-mode reparametrize returns mu
        check(o[0], r[0], '%s eval image logits %d' % (exp, c))
        check(o[1], r[1], '%s eval label logits %d' % (exp, c))
    check(o_total.item(), total.item(), exp + ' eval total')
    check([t.item() for t in o_terms], fx['terms'], exp + ' eval terms')
    meta = dict(exp=exp, batch=BATCH, n_latents=d, weight_seed=WEIGHT_SEED, input_seed=4321,
                lambda_image=1.0, lambda_label=LAMBDA_LABEL[exp] if exp == 'celeba' else 1.0)
    return fx, meta


def main():
    torch.set_num_threads(4)
    for exp in ('mnist', 'fashionmnist', 'celeba', 'celeba19'):
        fx, meta = run(exp)
        save('eval_' + exp, fx, meta)


if __name__ == '__main__':
    main()

"""Drop-in for the reference's ``celeba/train.py``: same CLI (--n-latents --batch-size --epochs
--annealing-epochs --lr --log-interval --lambda-image --lambda-attrs --cuda), same function names,
log lines and checkpoint format; the per-batch body is the fused HIP step."""
import os
import sys

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mvae_amd  # noqa: F401
    __package__ = 'multimodal-vae-public_amd.celeba'

from ..functional import binary_cross_entropy_with_logits  # noqa: E402,F401
from ..functional import elbo_loss_attrs as elbo_loss  # noqa: E402
from ..train_common import AverageMeter, add_extra_flags, make_load_checkpoint, run, save_checkpoint  # noqa: E402,F401
from .model import MVAE, N_ATTRS  # noqa: E402,F401

load_checkpoint = make_load_checkpoint(MVAE)


def _test_total(model, image, attrs, args):
    """celeba/train.py:232-245: the test pass uses the CLI lambdas, beta = 1."""
    kw = dict(lambda_image=args.lambda_image, lambda_attrs=args.lambda_attrs)
    r1 = model(image, attrs)
    r2 = model(image)
    r3 = model(attrs=attrs)
    return (elbo_loss(r1[0], image, r1[1], attrs, r1[2], r1[3], **kw)
            + elbo_loss(r2[0], image, None, None, r2[2], r2[3], **kw)
            + elbo_loss(None, None, r3[1], attrs, r3[2], r3[3], **kw))


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--n-latents', type=int, default=100,
                        help='size of the latent embedding [default: 100]')
    parser.add_argument('--batch-size', type=int, default=100, metavar='N',
                        help='input batch size for training [default: 100]')
    parser.add_argument('--epochs', type=int, default=100, metavar='N',
                        help='number of epochs to train [default: 100]')
    parser.add_argument('--annealing-epochs', type=int, default=20, metavar='N',
                        help='number of epochs to anneal KL for [default: 20]')
    parser.add_argument('--lr', type=float, default=1e-4, metavar='LR',
                        help='learning rate [default: 1e-4]')
    parser.add_argument('--log-interval', type=int, default=10, metavar='N',
                        help='how many batches to wait before logging training status [default: 10]')
    parser.add_argument('--lambda-image', type=float, default=1.,
                        help='multipler for image reconstruction [default: 1]')
    parser.add_argument('--lambda-attrs', type=float, default=10.,
                        help='multipler for attributes reconstruction [default: 10]')
    parser.add_argument('--cuda', action='store_true', default=False,
                        help='enables CUDA training [default: False]')
    add_extra_flags(parser)
    args = parser.parse_args()
    run('celeba', MVAE, _test_total, args, args.lambda_attrs)

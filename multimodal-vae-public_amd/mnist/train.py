"""Drop-in for the reference's ``mnist/train.py``: same CLI (--n-latents --batch-size --epochs
--annealing-epochs --lr --log-interval --lambda-image --lambda-text --cuda), same loss functions
by name, same log lines and checkpoint format; the per-batch body is the fused HIP step.

    python -m torch.distributed.run ... -m / or:  python multimodal-vae-public_amd/mnist/train.py --cuda --synthetic
"""
import os
import sys

if __package__ in (None, ''):      # executed as a script, like the reference (`python train.py`)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mvae_amd  # noqa: F401
    __package__ = 'multimodal-vae-public_amd.mnist'

from ..functional import binary_cross_entropy_with_logits, cross_entropy  # noqa: E402,F401
from ..functional import elbo_loss_label as elbo_loss  # noqa: E402
from ..train_common import AverageMeter, add_extra_flags, make_load_checkpoint, run, save_checkpoint  # noqa: E402,F401
from .model import MVAE  # noqa: E402

load_checkpoint = make_load_checkpoint(MVAE)


def _test_total(model, image, text, args):
    """The reference's test(): three calls, default lambdas and beta = 1 (mnist/train.py:242-249)."""
    r1 = model(image, text)
    r2 = model(image)
    r3 = model(text=text)
    return (elbo_loss(r1[0], image, r1[1], text, r1[2], r1[3])
            + elbo_loss(r2[0], image, None, None, r2[2], r2[3])
            + elbo_loss(None, None, r3[1], text, r3[2], r3[3]))


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--n-latents', type=int, default=64,
                        help='size of the latent embedding [default: 64]')
    parser.add_argument('--batch-size', type=int, default=100, metavar='N',
                        help='input batch size for training [default: 100]')
    parser.add_argument('--epochs', type=int, default=500, metavar='N',
                        help='number of epochs to train [default: 500]')
    parser.add_argument('--annealing-epochs', type=int, default=200, metavar='N',
                        help='number of epochs to anneal KL for [default: 200]')
    parser.add_argument('--lr', type=float, default=1e-3, metavar='LR',
                        help='learning rate [default: 1e-3]')
    parser.add_argument('--log-interval', type=int, default=10, metavar='N',
                        help='how many batches to wait before logging training status [default: 10]')
    parser.add_argument('--lambda-image', type=float, default=1.,
                        help='multipler for image reconstruction [default: 1]')
    parser.add_argument('--lambda-text', type=float, default=10.,
                        help='multipler for text reconstruction [default: 10]')
    parser.add_argument('--cuda', action='store_true', default=False,
                        help='enables CUDA training [default: False]')
    add_extra_flags(parser)
    args = parser.parse_args()
    run('mnist', MVAE, _test_total, args, args.lambda_text, annealing_epoch_offset=0)

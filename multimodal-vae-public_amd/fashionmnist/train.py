"""Drop-in for the reference's ``fashionmnist/train.py``: same CLI (--n-latents --batch-size --epochs
--annealing-epochs --lr --log-interval --lambda-image --lambda-text --cuda), same loss functions
by name, same log lines and checkpoint format; the per-batch body is the fused HIP step.

    python -m torch.distributed.run ... -m / or:  python multimodal-vae-public_amd/fashionmnist/train.py --cuda --synthetic
"""
import os
import sys

if __package__ in (None, ''):      # executed as a script, like the reference (`python train.py`)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mvae_amd  # noqa: F401
    __package__ = 'multimodal-vae-public_amd.fashionmnist'

from ..functional import binary_cross_entropy_with_logits, cross_entropy  # noqa: E402,F401
from ..functional import elbo_loss_label as elbo_loss  # noqa: E402
from ..train_common import AverageMeter, make_load_checkpoint, reference_parser, run, save_checkpoint  # noqa: E402,F401
from .model import MVAE  # noqa: E402

load_checkpoint = make_load_checkpoint(MVAE)


def _test_total(model, image, text, args):
    """The reference's test(): three calls, default lambdas and beta = 1 (fashionmnist/train.py:242-249)."""
    r1 = model(image, text)
    r2 = model(image)
    r3 = model(text=text)
    return (elbo_loss(r1[0], image, r1[1], text, r1[2], r1[3])
            + elbo_loss(r2[0], image, None, None, r2[2], r2[3])
            + elbo_loss(None, None, r3[1], text, r3[2], r3[3]))


if __name__ == "__main__":
    args = reference_parser('fashionmnist').parse_args()
    run('fashionmnist', MVAE, _test_total, args, args.lambda_text, annealing_epoch_offset=1)

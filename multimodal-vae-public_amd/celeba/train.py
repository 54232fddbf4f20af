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
from ..train_common import AverageMeter, make_load_checkpoint, reference_parser, run, save_checkpoint  # noqa: E402,F401
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
    args = reference_parser('celeba').parse_args()
    run('celeba', MVAE, _test_total, args, args.lambda_attrs)

"""Drop-in for the reference's ``celeba/sample.py``: same positional argument and flags
(model_path, --n-samples, --condition-on-image, --condition-on-text, --cuda), same four modes
(celeba/sample.py:86-116), outputs ``sample_image.png`` and ``sample_attrs.txt``.  See sample_common.py for the
two additions (--image-file / --synthetic) that stand in for the dataset draw."""
import os
import sys

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mvae_amd  # noqa: F401
    __package__ = 'multimodal-vae-public_amd.celeba'

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ..sample_common import (add_common_flags, generate, load_image, need_cuda, posterior,  # noqa: E402
                             save_image)
from .train import load_checkpoint  # noqa: E402
from .. import kernels as K  # noqa: E402

from ..sample_common import CELEBA_ATTRS  # noqa: E402


def tensor_to_attributes(tensor):
    """18 probabilities -> names of the attributes above 0.5 (celeba/datasets.py:138-152)."""
    return [CELEBA_ATTRS[i] for i in range(tensor.size(0)) if torch.round(tensor[i]) > 0.5]


def _attr_index(spec):
    return int(spec) if str(spec).lstrip('-').isdigit() else CELEBA_ATTRS.index(spec)


def fetch_celeba_image(spec, args):
    """An image to condition on: --image-file or a random image with --synthetic (the reference
    scans the CelebA test partition for one with attribute ``spec``, celeba/sample.py:20-46)."""
    if args.image_file:
        return load_image(args.image_file, (3, 64, 64))
    if args.synthetic:
        return torch.rand(1, 3, 64, 64)
    raise SystemExit('the CelebA dataset loader is out of scope: pass --image-file or --synthetic')


def fetch_celeba_attrs(spec):
    attrs = torch.zeros(len(CELEBA_ATTRS))
    attrs[_attr_index(spec)] = 1
    return attrs.unsqueeze(0)


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser()
    add_common_flags(parser)
    parser.add_argument('--condition-on-image', type=str, default=None,
                        help='If given, generate attributes conditioned on an image (attribute name/index).')
    parser.add_argument('--condition-on-text', type=str, default=None,
                        help='If given, generate images conditioned on an attribute (name or index).')
    args = parser.parse_args()
    need_cuda(args)
    model = load_checkpoint(args.model_path, use_cuda=True)
    model.cuda().eval()
    image = fetch_celeba_image(args.condition_on_image, args).cuda() if args.condition_on_image is not None else None
    attrs = fetch_celeba_attrs(args.condition_on_text).cuda() if args.condition_on_text is not None else None
    mu, std = posterior(model, image, attrs)
    _, image_recon, attr_logits = generate(model, args.n_samples, mu, std)
    save_image(image_recon.reshape(args.n_samples, 3, 64, 64), os.path.join(args.out_dir, 'sample_image.png'))
    attr_logits = attr_logits.contiguous()
    attrs_prob = torch.empty_like(attr_logits)
    K.sigmoid_fwd(attr_logits, attrs_prob)                    # F.sigmoid of celeba/sample.py:127 as a HIP launch
    attrs_recon = attrs_prob.cpu()
    with open(os.path.join(args.out_dir, 'sample_attrs.txt'), 'w') as fp:
        for i in range(attrs_recon.size(0)):
            fp.write('%s\n' % ','.join(tensor_to_attributes(attrs_recon[i])))

"""Drop-in for the reference's ``fashionmnist/sample.py``: same positional argument and flags
(model_path, --n-samples, --condition-on-image, --condition-on-text, --cuda), same four modes
(fashionmnist/sample.py:73-102), outputs ``sample_image.png`` and ``sample_text.txt``.  See sample_common.py for the
two additions (--image-file / --synthetic) that stand in for the dataset draw."""
import os
import sys

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mvae_amd  # noqa: F401
    __package__ = 'multimodal-vae-public_amd.fashionmnist'

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ..sample_common import (add_common_flags, generate, load_image, need_cuda, posterior,  # noqa: E402
                             save_image)
from .train import load_checkpoint  # noqa: E402

from ..sample_common import FASHION_LABELS as LABEL_IX_TO_STRING  # noqa: E402,F401


def fetch_fashionmnist_image(label, args):
    """An image of class ``label``: --image-file, a random image with --synthetic, else a random
    test-set image of that class (fashionmnist/sample.py:17-33; needs torchvision + the dataset)."""
    if args.image_file:
        return load_image(args.image_file, (1, 28, 28))
    if args.synthetic:
        return torch.rand(1, 1, 28, 28)
    try:
        from torchvision import datasets, transforms
    except ImportError:
        raise SystemExit('torchvision is not installed: pass --image-file or --synthetic')
    ds = datasets.FashionMNIST('./data', train=False, download=True, transform=transforms.ToTensor())
    images, labels = ds.data.numpy(), ds.targets.numpy()
    images = images[labels == label]
    image = images[np.random.choice(np.arange(images.shape[0]))]
    return torch.from_numpy(image).float().reshape(1, 1, 28, 28)     # raw 0..255, like the reference


def fetch_fashionmnist_text(label):
    return torch.LongTensor([label])


if __name__ == "__main__":
    import argparse
    parser = argparse.ArgumentParser()
    add_common_flags(parser)
    parser.add_argument('--condition-on-image', type=int, default=None,
                        help='If True, generate text conditioned on an image.')
    parser.add_argument('--condition-on-text', type=int, default=None,
                        help='If True, generate images conditioned on a text.')
    args = parser.parse_args()
    need_cuda(args)
    model = load_checkpoint(args.model_path, use_cuda=True)
    model.cuda().eval()
    # like the reference, the truthiness of the flags picks the mode (so class 0 means "not given")
    image = fetch_fashionmnist_image(args.condition_on_image, args).cuda() if args.condition_on_image else None
    text = fetch_fashionmnist_text(args.condition_on_text).cuda() if args.condition_on_text else None
    mu, std = posterior(model, image, text)
    _, img_recon, txt_logits = generate(model, args.n_samples, mu, std)
    save_image(img_recon.reshape(args.n_samples, 1, 28, 28), os.path.join(args.out_dir, 'sample_image.png'))
    txt = torch.log_softmax(txt_logits, dim=1).cpu().numpy().argmax(axis=1).tolist()
    with open(os.path.join(args.out_dir, 'sample_text.txt'), 'w') as fp:
        for i, item in enumerate(txt):
            fp.write('Text (%d): %s\n' % (i, LABEL_IX_TO_STRING[item]))

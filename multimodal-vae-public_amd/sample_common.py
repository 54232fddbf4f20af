"""Conditional generation shared by the ``sample.py`` drop-ins (mnist/sample.py:49-122,
fashionmnist/sample.py, celeba/sample.py:63-139): load a checkpoint, get the posterior of the
conditioning modalities with ``model.infer`` (eval mode: BatchNorm running statistics, no Dropout),
draw ``n_samples`` latents z = mu + std * eps, decode both modalities on the GPU and write
``sample_image.png`` + the label / attribute text file.

What differs from the reference and why:
  * the conditioning image comes from ``--image-file`` (a .npy / .pt array in [0, 1]) or, with
    ``--synthetic``, a random image -- the box has neither torchvision nor the datasets; with
    torchvision installed the MNIST test set is used exactly like mnist/sample.py:17-33;
  * ``save_image`` is a 60-line PNG grid writer with torchvision's default layout (8 per row,
    2-pixel padding) instead of ``torchvision.utils.save_image``;
  * celeba/sample.py reads ``args.condition_on_attrs`` and calls ``model.get_params`` -- neither
    exists in the reference (:87,98), so that script cannot run; here the flags it declares
    (--condition-on-image / --condition-on-text) select an attribute NAME or index and
    ``model.infer`` is used, which is what the MNIST script does.
"""
import struct
import zlib

import numpy as np
import torch

from . import kernels as K

# the 18 attributes celeba/datasets.py:34 keeps, in column order (names from the CelebA annotation
# header; indices 4,5,8,9,11,12,15,17,18,20,21,22,26,28,31,32,33,35 of the 40)
CELEBA_ATTRS = ['Bald', 'Bangs', 'Black_Hair', 'Blond_Hair', 'Brown_Hair', 'Bushy_Eyebrows', 'Eyeglasses',
                'Gray_Hair', 'Heavy_Makeup', 'Male', 'Mouth_Slightly_Open', 'Mustache', 'Pale_Skin',
                'Receding_Hairline', 'Smiling', 'Straight_Hair', 'Wavy_Hair', 'Wearing_Hat']
FASHION_LABELS = {0: 'T-shirt/top', 1: 'Trouser', 2: 'Pullover', 3: 'Dress', 4: 'Coat', 5: 'Sandal',
                  6: 'Shirt', 7: 'Sneaker', 8: 'Bag', 9: 'Ankle boot'}


# ----------------------------------------------------------------------------- PNG grid
def make_grid(images, nrow=8, padding=2):
    """[N, C, H, W] in [0, 1] -> uint8 [H', W', 3] laid out like torchvision.utils.make_grid."""
    x = np.asarray(images, dtype=np.float32)
    if x.ndim != 4:
        raise ValueError('expected [N, C, H, W]')
    n, c, h, w = x.shape
    if c == 1:
        x = np.repeat(x, 3, axis=1)
    elif c != 3:
        raise ValueError('images must have 1 or 3 channels')
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    H, W = h + padding, w + padding
    grid = np.zeros((3, H * ymaps + padding, W * xmaps + padding), dtype=np.float32)
    for k in range(n):
        r, col = divmod(k, xmaps)
        grid[:, r * H + padding:r * H + padding + h, col * W + padding:col * W + padding + w] = x[k]
    return (np.clip(grid * 255.0 + 0.5, 0, 255)).astype(np.uint8).transpose(1, 2, 0)


def write_png(path, rgb):
    """uint8 [H, W, 3] -> 8-bit RGB PNG (zlib + CRC only)."""
    h, w, _ = rgb.shape
    raw = b''.join(b'\x00' + rgb[r].tobytes() for r in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0))
                + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def save_image(tensor, path, nrow=8, padding=2):
    write_png(path, make_grid(tensor.detach().cpu().numpy(), nrow=nrow, padding=padding))


# ----------------------------------------------------------------------------- generation
def posterior(model, image=None, label=None):
    """(mu, std) of the conditioning modalities; the prior N(0, 1) when there are none
    (mnist/sample.py:71-100)."""
    dev = next(model.parameters()).device
    if image is None and label is None:
        return torch.zeros(1, device=dev), torch.ones(1, device=dev)
    with torch.no_grad():
        if model.KIND in ('celeba', 'celeba19'):
            mu, logvar = model.infer(image=image, attrs=label)
        else:
            mu, logvar = model.infer(image=image, text=label)
    # std = exp(logvar / 2) (mnist/sample.py:100): the reparameterisation kernel at mu = 0, eps = 1
    mu, logvar = mu.contiguous(), logvar.contiguous()
    std = torch.empty_like(logvar)
    K.reparam_fwd(torch.zeros_like(logvar), logvar, torch.ones_like(logvar), std)
    return mu, std


def generate(model, n_samples, mu, std, eps=None):
    """z = eps * std + mu for ``n_samples`` draws (mnist/sample.py:102-109), decoded by both
    decoders.  Returns (z, image probabilities, label logits)."""
    dev = next(model.parameters()).device
    if eps is None:
        eps = torch.randn(n_samples, model.n_latents)
    eps = eps.to(dev).float().contiguous()
    z = torch.empty_like(eps)
    K.affine_fwd(eps, std.reshape(-1).contiguous(), mu.reshape(-1).contiguous(), z)     # one posterior row, n draws
    with torch.no_grad():
        logits = model.image_decoder(z).contiguous()
        img = torch.empty_like(logits)
        K.sigmoid_fwd(logits, img)                                                     # F.sigmoid, mnist/sample.py:111
        lbl = model.label_decoder(z)
    return z, img, lbl


def load_image(path, shape):
    arr = torch.load(path) if path.endswith('.pt') else torch.from_numpy(np.load(path))
    return arr.float().reshape((1,) + tuple(shape))


def add_common_flags(parser):
    parser.add_argument('model_path', type=str, help='path to trained model file')
    parser.add_argument('--n-samples', type=int, default=64,
                        help='Number of images and texts to sample [default: 64]')
    parser.add_argument('--cuda', action='store_true', default=False,
                        help='enables CUDA training [default: False]')
    parser.add_argument('--image-file', type=str, default=None,
                        help='.npy / .pt image in [0, 1] to condition on (instead of a dataset draw)')
    parser.add_argument('--synthetic', action='store_true', default=False,
                        help='condition on a random-pixel image when no dataset is available')
    parser.add_argument('--out-dir', type=str, default='.')


def need_cuda(args):
    args.cuda = args.cuda and torch.cuda.is_available()
    if not args.cuda:
        raise SystemExit('this drop-in decodes with HIP kernels: pass --cuda on a ROCm GPU box')

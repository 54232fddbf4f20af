"""FashionMNIST MVAE on HIP -- drop-in for the reference's ``fashionmnist/model.py``.

    MVAE          fashionmnist/model.py:18-68
    ImageEncoder  fashionmnist/model.py:70-94    Conv(1,64,4,2,1) Swish Conv(64,128,4,2,1) Swish
                                                 | Linear(6272,512) Swish Linear(512,2D)
    ImageDecoder  fashionmnist/model.py:97-121   Linear(D,512) Swish Linear(512,6272) Swish
                                                 | ConvT(128,64,4,2,1) Swish ConvT(64,1,4,2,1)
    TextEncoder   fashionmnist/model.py:124-143  Embedding(10,512) Swish Linear Swish Linear(512,2D)
    TextDecoder   fashionmnist/model.py:146-165
    ProductOfExperts variant A                   fashionmnist/model.py:175-182
"""
import torch.nn as nn

from .. import layers as L
from ..base import MVAEBase, Stack
# module-level names of the reference's model.py (``from model import ProductOfExperts, Swish, prior_expert``):
# ProductOfExperts here is variant A -- fashionmnist/model.py:168-182
from ..base import ProductOfExperts, prior_expert  # noqa: F401
from ..layers import Swish  # noqa: F401


class _SplitEncoder(Stack):
    def heads(self, x):
        return self.run(x)

    def forward(self, x):
        h = self.heads(x)
        return h[:, :self.n_latents], h[:, self.n_latents:]


class ImageEncoder(_SplitEncoder):
    def __init__(self, n_latents):
        super().__init__()
        self.features = nn.Sequential(
            L.Conv2d(1, 64, 4, 2, 1, bias=False), L.Swish(),
            L.Conv2d(64, 128, 4, 2, 1, bias=False), L.Swish())
        self.classifier = nn.Sequential(
            L.Linear(128 * 7 * 7, 512), L.Swish(), L.Linear(512, n_latents * 2))
        self.n_latents = n_latents

    def stack_modules(self):
        return [self.features, L.View(128 * 7 * 7), self.classifier]


class ImageDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.n_latents = n_latents
        self.upsampler = nn.Sequential(
            L.Linear(n_latents, 512), L.Swish(), L.Linear(512, 128 * 7 * 7), L.Swish())
        self.hallucinate = nn.Sequential(
            L.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), L.Swish(),
            L.ConvTranspose2d(64, 1, 4, 2, 1, bias=False))

    def stack_modules(self):
        return [self.upsampler, L.View(128, 7, 7), self.hallucinate]

    def forward(self, z):
        return self.run(z)  # NOTE: logits, no sigmoid


class TextEncoder(_SplitEncoder):
    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            L.Embedding(10, 512), L.Swish(), L.Linear(512, 512), L.Swish(),
            L.Linear(512, n_latents * 2))
        self.n_latents = n_latents

    def stack_modules(self):
        return [self.net]


class TextDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            L.Linear(n_latents, 512), L.Swish(), L.Linear(512, 512), L.Swish(),
            L.Linear(512, 512), L.Swish(), L.Linear(512, 10))

    def stack_modules(self):
        return [self.net]

    def forward(self, z):
        return self.run(z)  # NOTE: logits, no softmax


class MVAE(MVAEBase):
    POE_VARIANT = 'A'
    KIND = 'fashionmnist'
    LABEL_KIND = 'class'
    HAS_BN = False
    IMAGE_SHAPE = (1, 28, 28)

    def __init__(self, n_latents):
        super().__init__(n_latents)
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.text_encoder = TextEncoder(n_latents)
        self.text_decoder = TextDecoder(n_latents)

    label_encoder = property(lambda self: self.text_encoder)
    label_decoder = property(lambda self: self.text_decoder)

    def arena_order(self):
        return [self.image_decoder, self.text_decoder, self.text_encoder, self.image_encoder]

    def arena_tail(self):
        return [self.image_encoder.features]

    def forward(self, image=None, text=None, eps=None):
        mu, logvar, z = self._infer(image, text, eps, want_z=True)
        return self.image_decoder(z), self.text_decoder(z), mu, logvar

    def infer(self, image=None, text=None):
        mu, logvar, _ = self._infer(image, text, None, want_z=False)
        return mu, logvar

    def _infer(self, image, text, eps, want_z):
        self.finalize()
        heads = []
        if image is not None:
            heads.append(self.image_encoder.heads(image))
        if text is not None:
            heads.append(self.text_encoder.heads(text))
        if not heads:
            raise ValueError('at least one modality is required')
        return self._fuse(heads, eps, want_z)

"""CelebA MVAE (image + 18 attributes) on HIP -- drop-in for the reference's ``celeba/model.py``.

    MVAE              celeba/model.py:13-63
    ImageEncoder      celeba/model.py:66-100    DCGAN conv stack with BatchNorm2d + Dropout(0.1)
    ImageDecoder      celeba/model.py:103-133   Linear(D,6400) + ConvTranspose stack with BatchNorm2d
    AttributeEncoder  celeba/model.py:136-160   18 -> 512 -> 512 -> 2D with BatchNorm1d
    AttributeDecoder  celeba/model.py:163-190   D -> 512 x3 -> 18 with BatchNorm1d
    ProductOfExperts variant B                  celeba/model.py:200-207

The image stacks are shared with CelebA-19 (celeba19/model.py:92-159 is the same text).
"""
import torch.nn as nn

from .. import layers as L
from ..base import MVAEBase, Stack
# module-level names of the reference's model.py (``from model import ProductOfExperts, Swish, prior_expert``):
# ProductOfExperts here is variant B -- celeba/model.py:193-207
from ..base import ProductOfExpertsB as ProductOfExperts, prior_expert  # noqa: F401
from ..layers import Swish  # noqa: F401

N_ATTRS = 18  # celeba/datasets.py:34


class ImageEncoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.features = nn.Sequential(
            L.Conv2d(3, 32, 4, 2, 1, bias=False), L.Swish(),
            L.Conv2d(32, 64, 4, 2, 1, bias=False), L.BatchNorm2d(64), L.Swish(),
            L.Conv2d(64, 128, 4, 2, 1, bias=False), L.BatchNorm2d(128), L.Swish(),
            L.Conv2d(128, 256, 4, 1, 0, bias=False), L.BatchNorm2d(256), L.Swish())
        self.classifier = nn.Sequential(
            L.Linear(256 * 5 * 5, 512), L.Swish(), L.Dropout(p=0.1), L.Linear(512, n_latents * 2))
        self.n_latents = n_latents

    def stack_modules(self):
        return [self.features, L.View(256 * 5 * 5), self.classifier]

    # the train step runs this encoder once per batch and fans the Dropout draw out (engine.py)
    def trunk_modules(self):
        return [self.features, L.View(256 * 5 * 5), self.classifier[0], self.classifier[1]]

    def head_modules(self):
        return [self.classifier[3]]

    def heads(self, x, dropout_mask=None):
        masks = None
        if self.training:
            if dropout_mask is None:
                dropout_mask = _owner_mvae(self).device_bernoulli(0.9, x.shape[0], 512)
            masks = [dropout_mask.contiguous().float()]
        return self.run(x, masks=masks)

    def forward(self, x, dropout_mask=None):
        h = self.heads(x, dropout_mask)
        return h[:, :self.n_latents], h[:, self.n_latents:]


def _owner_mvae(module):
    owner = module.__dict__.get('_owner')
    if owner is None:
        raise RuntimeError('encoder is not attached to an MVAE (needed for device-side dropout noise); '
                           'pass dropout_mask explicitly')
    return owner


class ImageDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.upsample = nn.Sequential(L.Linear(n_latents, 256 * 5 * 5), L.Swish())
        self.hallucinate = nn.Sequential(
            L.ConvTranspose2d(256, 128, 4, 1, 0, bias=False), L.BatchNorm2d(128), L.Swish(),
            L.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), L.BatchNorm2d(64), L.Swish(),
            L.ConvTranspose2d(64, 32, 4, 2, 1, bias=False), L.BatchNorm2d(32), L.Swish(),
            L.ConvTranspose2d(32, 3, 4, 2, 1, bias=False))

    def stack_modules(self):
        return [self.upsample, L.View(256, 5, 5), self.hallucinate]

    def forward(self, z):
        return self.run(z)  # NOTE: logits, no sigmoid


class AttributeEncoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            L.Linear(N_ATTRS, 512), L.BatchNorm1d(512), L.Swish(),
            L.Linear(512, 512), L.BatchNorm1d(512), L.Swish(),
            L.Linear(512, n_latents * 2))
        self.n_latents = n_latents

    def stack_modules(self):
        return [self.net]

    def heads(self, x):
        return self.run(x.float().contiguous())

    def forward(self, x):
        h = self.heads(x)
        return h[:, :self.n_latents], h[:, self.n_latents:]


class AttributeDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            L.Linear(n_latents, 512), L.BatchNorm1d(512), L.Swish(),
            L.Linear(512, 512), L.BatchNorm1d(512), L.Swish(),
            L.Linear(512, 512), L.BatchNorm1d(512), L.Swish(),
            L.Linear(512, N_ATTRS))

    def stack_modules(self):
        return [self.net]

    def forward(self, z):
        return self.run(z)  # NOTE: logits, no sigmoid


class MVAE(MVAEBase):
    POE_VARIANT = 'B'
    KIND = 'celeba'
    LABEL_KIND = 'attrs'
    HAS_BN = True
    IMAGE_SHAPE = (3, 64, 64)

    def __init__(self, n_latents):
        super().__init__(n_latents)
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.attrs_encoder = AttributeEncoder(n_latents)
        self.attrs_decoder = AttributeDecoder(n_latents)
        self.image_encoder.__dict__['_owner'] = self

    label_encoder = property(lambda self: self.attrs_encoder)
    label_decoder = property(lambda self: self.attrs_decoder)

    def arena_order(self):
        return [self.image_decoder, self.attrs_decoder, self.attrs_encoder, self.image_encoder]

    def arena_tail(self):
        return [self.image_encoder.features]

    def forward(self, image=None, attrs=None, eps=None, dropout_mask=None):
        mu, logvar, z = self._infer(image, attrs, eps, dropout_mask, want_z=True)
        return self.image_decoder(z), self.attrs_decoder(z), mu, logvar

    def infer(self, image=None, attrs=None):
        mu, logvar, _ = self._infer(image, attrs, None, None, want_z=False)
        return mu, logvar

    def _infer(self, image, attrs, eps, dropout_mask, want_z):
        self.finalize()
        heads = []
        if image is not None:
            heads.append(self.image_encoder.heads(image, dropout_mask))
        if attrs is not None:
            heads.append(self.attrs_encoder.heads(attrs))
        if not heads:
            raise ValueError('at least one modality is required')
        return self._fuse(heads, eps, want_z)

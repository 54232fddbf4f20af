"""CelebA-19 MVAE (image + 18 separate attribute experts) on HIP -- drop-in for the reference's
``celeba19/model.py``.

    MVAE              celeba19/model.py:15-89    19 modalities + prior = up to 20 experts in the PoE
    ImageEncoder      celeba19/model.py:92-126   (same text as celeba/model.py:66-100)
    ImageDecoder      celeba19/model.py:129-159
    AttributeEncoder  celeba19/model.py:162-184  x.long() -> Embedding(2,512) Swish Linear Swish Linear(512,2D)
    AttributeDecoder  celeba19/model.py:187-209  D -> 512 x3 -> 1
    ProductOfExperts variant B                   celeba19/model.py:219-226
"""
import torch
import torch.nn as nn

from .. import layers as L
from ..base import MVAEBase, Stack
# module-level names of the reference's model.py (``from model import ProductOfExperts, Swish, prior_expert``):
# ProductOfExperts here is variant B -- celeba19/model.py:212-226
from ..base import ProductOfExpertsB as ProductOfExperts, prior_expert  # noqa: F401
from ..layers import Swish  # noqa: F401
from ..celeba.model import ImageDecoder, ImageEncoder, N_ATTRS  # noqa: F401  (identical stacks)


class AttributeEncoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            L.Embedding(2, 512), L.Swish(), L.Linear(512, 512), L.Swish(),
            L.Linear(512, n_latents * 2))
        self.n_latents = n_latents

    def stack_modules(self):
        return [self.net]

    def heads(self, x):
        # the reference casts with .long() (twice: celeba19/model.py:83,183); the gather kernel
        # reads the {0,1} floats directly
        if x.dtype not in (torch.int64, torch.float32):
            x = x.float()
        return self.run(x.contiguous())

    def forward(self, x):
        h = self.heads(x)
        return h[:, :self.n_latents], h[:, self.n_latents:]


class AttributeDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.net = nn.Sequential(
            L.Linear(n_latents, 512), L.Swish(), L.Linear(512, 512), L.Swish(),
            L.Linear(512, 512), L.Swish(), L.Linear(512, 1))

    def stack_modules(self):
        return [self.net]

    def forward(self, z):
        return self.run(z)  # [B, 1] logits, no sigmoid


class MVAE(MVAEBase):
    POE_VARIANT = 'B'
    KIND = 'celeba19'
    HAS_BN = True
    IMAGE_SHAPE = (3, 64, 64)

    def __init__(self, n_latents):
        super().__init__(n_latents)
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.attr_encoders = nn.ModuleList([AttributeEncoder(n_latents) for _ in range(N_ATTRS)])
        self.attr_decoders = nn.ModuleList([AttributeDecoder(n_latents) for _ in range(N_ATTRS)])
        self.image_encoder.__dict__['_owner'] = self

    def arena_order(self):
        return ([self.image_decoder] + list(self.attr_decoders) + list(self.attr_encoders)
                + [self.image_encoder])

    def arena_tail(self):
        return [self.image_encoder.features]

    def forward(self, image=None, attrs=None, eps=None, dropout_mask=None):
        """``attrs``: list of 18 tensors [B] (or None for a missing attribute), like the
        reference (celeba19/model.py:41-61).  Returns (image_recon, [18 x [B]], mu, logvar)."""
        mu, logvar, z = self._infer(image, attrs, eps, dropout_mask, want_z=True)
        image_recon = self.image_decoder(z)
        attr_recons = [self.attr_decoders[i](z).squeeze(1) for i in range(N_ATTRS)]
        return image_recon, attr_recons, mu, logvar

    def infer(self, image=None, attrs=None):
        mu, logvar, _ = self._infer(image, attrs, None, None, want_z=False)
        return mu, logvar

    def _infer(self, image, attrs, eps, dropout_mask, want_z):
        self.finalize()
        attrs = attrs if attrs is not None else [None] * N_ATTRS
        heads = []
        if image is not None:
            heads.append(self.image_encoder.heads(image, dropout_mask))
        for i in range(N_ATTRS):
            if attrs[i] is not None:
                heads.append(self.attr_encoders[i].heads(attrs[i]))
        if not heads:
            raise ValueError('at least one modality is required')
        return self._fuse(heads, eps, want_z)

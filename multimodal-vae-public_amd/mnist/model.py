"""MNIST MVAE on HIP -- drop-in for the reference's ``mnist/model.py`` (same class names,
constructor arguments, ``forward`` keywords, return tuple and ``state_dict`` keys).

    MVAE            mnist/model.py:14-64
    ImageEncoder    mnist/model.py:67-84     784 -> 512 -> 512 -> (D, D)
    ImageDecoder    mnist/model.py:87-105    D -> 512 -> 512 -> 512 -> 784 (logits)
    TextEncoder     mnist/model.py:108-125   Embedding(10,512) -> 512 -> (D, D)
    TextDecoder     mnist/model.py:128-146   D -> 512 -> 512 -> 512 -> 10 (logits)
    ProductOfExperts variant A               mnist/model.py:156-163
"""
from .. import layers as L
from ..base import MVAEBase, Stack
# module-level names of the reference's model.py (``from model import ProductOfExperts, Swish, prior_expert``):
# ProductOfExperts here is variant A -- mnist/model.py:149-163
from ..base import ProductOfExperts, prior_expert  # noqa: F401
from ..layers import Swish  # noqa: F401


class _TwoHeadEncoder(Stack):
    """fc31 / fc32 stay separate parameters (reference names) but run as one GEMM of width 2D."""
    def heads(self, x):
        return self.run(x)

    def forward(self, x):
        h = self.heads(x)
        d = self.fc31.out_features
        return h[:, :d], h[:, d:]


class ImageEncoder(_TwoHeadEncoder):
    def __init__(self, n_latents):
        super().__init__()
        self.fc1 = L.Linear(784, 512)
        self.fc2 = L.Linear(512, 512)
        self.fc31 = L.Linear(512, n_latents)
        self.fc32 = L.Linear(512, n_latents)
        self.swish = L.Swish()

    def stack_modules(self):
        return [L.View(784), self.fc1, self.swish, self.fc2, self.swish, L.HeadPair(self.fc31, self.fc32)]


class ImageDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.fc1 = L.Linear(n_latents, 512)
        self.fc2 = L.Linear(512, 512)
        self.fc3 = L.Linear(512, 512)
        self.fc4 = L.Linear(512, 784)
        self.swish = L.Swish()

    def stack_modules(self):
        return [self.fc1, self.swish, self.fc2, self.swish, self.fc3, self.swish, self.fc4]

    def forward(self, z):
        return self.run(z)  # NOTE: logits, no sigmoid (reference :105)


class TextEncoder(_TwoHeadEncoder):
    def __init__(self, n_latents):
        super().__init__()
        self.fc1 = L.Embedding(10, 512)
        self.fc2 = L.Linear(512, 512)
        self.fc31 = L.Linear(512, n_latents)
        self.fc32 = L.Linear(512, n_latents)
        self.swish = L.Swish()

    def stack_modules(self):
        return [self.fc1, self.swish, self.fc2, self.swish, L.HeadPair(self.fc31, self.fc32)]


class TextDecoder(Stack):
    def __init__(self, n_latents):
        super().__init__()
        self.fc1 = L.Linear(n_latents, 512)
        self.fc2 = L.Linear(512, 512)
        self.fc3 = L.Linear(512, 512)
        self.fc4 = L.Linear(512, 10)
        self.swish = L.Swish()

    def stack_modules(self):
        return [self.fc1, self.swish, self.fc2, self.swish, self.fc3, self.swish, self.fc4]

    def forward(self, z):
        return self.run(z)  # NOTE: logits, no softmax (reference :146)


class MVAE(MVAEBase):
    """Multimodal Variational Autoencoder (image + digit label)."""
    POE_VARIANT = 'A'
    KIND = 'mnist'
    LABEL_KIND = 'class'
    HAS_BN = False
    IMAGE_SHAPE = (1, 28, 28)

    def __init__(self, n_latents):
        super().__init__(n_latents)
        self.image_encoder = ImageEncoder(n_latents)
        self.image_decoder = ImageDecoder(n_latents)
        self.text_encoder = TextEncoder(n_latents)
        self.text_decoder = TextDecoder(n_latents)

    # engine hooks
    label_encoder = property(lambda self: self.text_encoder)
    label_decoder = property(lambda self: self.text_decoder)

    def arena_order(self):
        return [self.image_decoder, self.text_decoder, self.text_encoder, self.image_encoder]

    def arena_tail(self):
        return [self.image_encoder.fc1]

    def arena_adjacent(self):
        out = []
        for enc in (self.image_encoder, self.text_encoder):
            out.append((enc.fc31.weight, enc.fc32.weight))
            out.append((enc.fc31.bias, enc.fc32.bias))
        return out

    def forward(self, image=None, text=None, eps=None):
        mu, logvar, z = self._infer(image, text, eps, want_z=True)
        img_recon = self.image_decoder(z)
        txt_recon = self.text_decoder(z)
        return img_recon, txt_recon, mu, logvar

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

"""Oracle (test infrastructure): the four hot-path MVAE architectures on torch CPU.

Own code; keeps the reference's ``state_dict`` keys so
that weights interchange with the reference modules and with the HIP modules.
Noise is explicit: ``forward(..., eps=, dropout_mask=)``; when a training-mode
call omits it, it is drawn from the global CPU generator in the reference's
order (dropout mask first -- only when the image encoder runs and has a
Dropout -- then the reparameterisation eps; SURVEY.md section 7 "identical seeds").
"""
import torch
import torch.nn as nn

from . import functional as OF

N_ATTRS = 18  # celeba/datasets.py:34


class Swish(nn.Module):
    def forward(self, x):
        return OF.swish(x)


class _MaskedDropout(nn.Module):
    """nn.Dropout(p) with the Bernoulli(1-p) keep-mask as an explicit input
    (celeba/model.py:91).  Parameter-free, so Sequential indices are unchanged."""
    def __init__(self, p):
        super().__init__()
        self.p = p
        self.mask = None  # set by the owner right before the call

    def forward(self, x):
        if not self.training:
            return x
        m = self.mask
        if m is None:
            m = torch.empty_like(x).bernoulli_(1 - self.p)
        self.mask = None
        return x * (m / (1 - self.p))


def _split(x, d):
    return x[:, :d], x[:, d:]


# ----------------------------------------------------------------------------
# MNIST -- mnist/model.py
# ----------------------------------------------------------------------------
class MnistImageEncoder(nn.Module):   # mnist/model.py:67-84
    def __init__(self, d):
        super().__init__()
        self.fc1 = nn.Linear(784, 512)
        self.fc2 = nn.Linear(512, 512)
        self.fc31 = nn.Linear(512, d)
        self.fc32 = nn.Linear(512, d)

    def forward(self, x):
        h = OF.swish(self.fc1(x.reshape(-1, 784)))
        h = OF.swish(self.fc2(h))
        return self.fc31(h), self.fc32(h)


class MnistImageDecoder(nn.Module):   # mnist/model.py:87-105
    def __init__(self, d):
        super().__init__()
        self.fc1 = nn.Linear(d, 512)
        self.fc2 = nn.Linear(512, 512)
        self.fc3 = nn.Linear(512, 512)
        self.fc4 = nn.Linear(512, 784)

    def forward(self, z):
        h = OF.swish(self.fc1(z))
        h = OF.swish(self.fc2(h))
        h = OF.swish(self.fc3(h))
        return self.fc4(h)


class MnistTextEncoder(nn.Module):    # mnist/model.py:108-125
    def __init__(self, d):
        super().__init__()
        self.fc1 = nn.Embedding(10, 512)
        self.fc2 = nn.Linear(512, 512)
        self.fc31 = nn.Linear(512, d)
        self.fc32 = nn.Linear(512, d)

    def forward(self, x):
        h = OF.swish(self.fc1(x))
        h = OF.swish(self.fc2(h))
        return self.fc31(h), self.fc32(h)


class MnistTextDecoder(nn.Module):    # mnist/model.py:128-146
    def __init__(self, d):
        super().__init__()
        self.fc1 = nn.Linear(d, 512)
        self.fc2 = nn.Linear(512, 512)
        self.fc3 = nn.Linear(512, 512)
        self.fc4 = nn.Linear(512, 10)

    def forward(self, z):
        h = OF.swish(self.fc1(z))
        h = OF.swish(self.fc2(h))
        h = OF.swish(self.fc3(h))
        return self.fc4(h)


class _BimodalMVAE(nn.Module):
    """MVAE.forward / infer / reparametrize shared by mnist, fashionmnist, celeba --
    mnist/model.py:29-64, celeba/model.py:28-63."""
    POE_VARIANT = 'A'
    HAS_DROPOUT = False

    def _label_encoder(self):
        raise NotImplementedError

    def infer(self, image=None, label=None, dropout_mask=None):
        mus, lvs = [], []
        if image is not None:
            if self.HAS_DROPOUT:
                self.image_encoder.classifier[2].mask = dropout_mask
            m, v = self.image_encoder(image)
            mus.append(m); lvs.append(v)
        if label is not None:
            m, v = self._label_encoder()(label)
            mus.append(m); lvs.append(v)
        return OF.poe_with_prior(mus, lvs, self.POE_VARIANT)

    def forward(self, image=None, label=None, eps=None, dropout_mask=None):
        mu, logvar = self.infer(image, label, dropout_mask)
        if self.training and eps is None:
            eps = torch.empty_like(mu).normal_()
        z = OF.reparametrize(mu, logvar, eps if self.training else None)
        return self._decode(z) + (mu, logvar, z)


class MnistMVAE(_BimodalMVAE):        # mnist/model.py:14-64
    POE_VARIANT = 'A'

    def __init__(self, n_latents):
        super().__init__()
        self.image_encoder = MnistImageEncoder(n_latents)
        self.image_decoder = MnistImageDecoder(n_latents)
        self.text_encoder = MnistTextEncoder(n_latents)
        self.text_decoder = MnistTextDecoder(n_latents)
        self.n_latents = n_latents

    def _label_encoder(self):
        return self.text_encoder

    def _decode(self, z):
        return self.image_decoder(z), self.text_decoder(z)


# ----------------------------------------------------------------------------
# FashionMNIST -- fashionmnist/model.py
# ----------------------------------------------------------------------------
class FmnistImageEncoder(nn.Module):  # fashionmnist/model.py:70-94
    def __init__(self, d):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(1, 64, 4, 2, 1, bias=False), Swish(),
            nn.Conv2d(64, 128, 4, 2, 1, bias=False), Swish())
        self.classifier = nn.Sequential(
            nn.Linear(128 * 7 * 7, 512), Swish(), nn.Linear(512, d * 2))
        self.n_latents = d

    def forward(self, x):
        x = self.features(x)
        x = self.classifier(x.reshape(x.size(0), -1))
        return _split(x, self.n_latents)


class FmnistImageDecoder(nn.Module):  # fashionmnist/model.py:97-121
    def __init__(self, d):
        super().__init__()
        self.upsampler = nn.Sequential(
            nn.Linear(d, 512), Swish(), nn.Linear(512, 128 * 7 * 7), Swish())
        self.hallucinate = nn.Sequential(
            nn.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), Swish(),
            nn.ConvTranspose2d(64, 1, 4, 2, 1, bias=False))

    def forward(self, z):
        z = self.upsampler(z)
        return self.hallucinate(z.reshape(-1, 128, 7, 7))


class FmnistTextEncoder(nn.Module):   # fashionmnist/model.py:124-143
    def __init__(self, d):
        super().__init__()
        self.net = nn.Sequential(
            nn.Embedding(10, 512), Swish(), nn.Linear(512, 512), Swish(),
            nn.Linear(512, d * 2))
        self.n_latents = d

    def forward(self, x):
        return _split(self.net(x), self.n_latents)


class FmnistTextDecoder(nn.Module):   # fashionmnist/model.py:146-165
    def __init__(self, d):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(d, 512), Swish(), nn.Linear(512, 512), Swish(),
            nn.Linear(512, 512), Swish(), nn.Linear(512, 10))

    def forward(self, z):
        return self.net(z)


class FmnistMVAE(_BimodalMVAE):       # fashionmnist/model.py:18-68
    POE_VARIANT = 'A'

    def __init__(self, n_latents):
        super().__init__()
        self.image_encoder = FmnistImageEncoder(n_latents)
        self.image_decoder = FmnistImageDecoder(n_latents)
        self.text_encoder = FmnistTextEncoder(n_latents)
        self.text_decoder = FmnistTextDecoder(n_latents)
        self.n_latents = n_latents

    def _label_encoder(self):
        return self.text_encoder

    def _decode(self, z):
        return self.image_decoder(z), self.text_decoder(z)


# ----------------------------------------------------------------------------
# CelebA -- celeba/model.py  (image stacks shared with celeba19/model.py)
# ----------------------------------------------------------------------------
class CelebaImageEncoder(nn.Module):  # celeba/model.py:66-100, celeba19/model.py:92-126
    def __init__(self, d):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 32, 4, 2, 1, bias=False), Swish(),
            nn.Conv2d(32, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.Conv2d(64, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.Conv2d(128, 256, 4, 1, 0, bias=False), nn.BatchNorm2d(256), Swish())
        self.classifier = nn.Sequential(
            nn.Linear(256 * 5 * 5, 512), Swish(), _MaskedDropout(0.1),
            nn.Linear(512, d * 2))
        self.n_latents = d

    def forward(self, x):
        x = self.features(x)
        x = self.classifier(x.reshape(-1, 256 * 5 * 5))
        return _split(x, self.n_latents)


class CelebaImageDecoder(nn.Module):  # celeba/model.py:103-133, celeba19/model.py:129-159
    def __init__(self, d):
        super().__init__()
        self.upsample = nn.Sequential(nn.Linear(d, 256 * 5 * 5), Swish())
        self.hallucinate = nn.Sequential(
            nn.ConvTranspose2d(256, 128, 4, 1, 0, bias=False), nn.BatchNorm2d(128), Swish(),
            nn.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), nn.BatchNorm2d(64), Swish(),
            nn.ConvTranspose2d(64, 32, 4, 2, 1, bias=False), nn.BatchNorm2d(32), Swish(),
            nn.ConvTranspose2d(32, 3, 4, 2, 1, bias=False))

    def forward(self, z):
        z = self.upsample(z)
        return self.hallucinate(z.reshape(-1, 256, 5, 5))


class CelebaAttrsEncoder(nn.Module):  # celeba/model.py:136-160
    def __init__(self, d):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(N_ATTRS, 512), nn.BatchNorm1d(512), Swish(),
            nn.Linear(512, 512), nn.BatchNorm1d(512), Swish(),
            nn.Linear(512, d * 2))
        self.n_latents = d

    def forward(self, x):
        return _split(self.net(x), self.n_latents)


class CelebaAttrsDecoder(nn.Module):  # celeba/model.py:163-190
    def __init__(self, d):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(d, 512), nn.BatchNorm1d(512), Swish(),
            nn.Linear(512, 512), nn.BatchNorm1d(512), Swish(),
            nn.Linear(512, 512), nn.BatchNorm1d(512), Swish(),
            nn.Linear(512, N_ATTRS))

    def forward(self, z):
        return self.net(z)


class CelebaMVAE(_BimodalMVAE):       # celeba/model.py:13-63
    POE_VARIANT = 'B'
    HAS_DROPOUT = True

    def __init__(self, n_latents):
        super().__init__()
        self.image_encoder = CelebaImageEncoder(n_latents)
        self.image_decoder = CelebaImageDecoder(n_latents)
        self.attrs_encoder = CelebaAttrsEncoder(n_latents)
        self.attrs_decoder = CelebaAttrsDecoder(n_latents)
        self.n_latents = n_latents

    def _label_encoder(self):
        return self.attrs_encoder

    def _decode(self, z):
        return self.image_decoder(z), self.attrs_decoder(z)


# ----------------------------------------------------------------------------
# CelebA-19 -- celeba19/model.py
# ----------------------------------------------------------------------------
class Celeba19AttrEncoder(nn.Module):  # celeba19/model.py:162-184
    def __init__(self, d):
        super().__init__()
        self.net = nn.Sequential(
            nn.Embedding(2, 512), Swish(), nn.Linear(512, 512), Swish(),
            nn.Linear(512, d * 2))
        self.n_latents = d

    def forward(self, x):
        return _split(self.net(x.long()), self.n_latents)


class Celeba19AttrDecoder(nn.Module):  # celeba19/model.py:187-209
    def __init__(self, d):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(d, 512), Swish(), nn.Linear(512, 512), Swish(),
            nn.Linear(512, 512), Swish(), nn.Linear(512, 1))

    def forward(self, z):
        return self.net(z)


class Celeba19MVAE(nn.Module):         # celeba19/model.py:15-89
    POE_VARIANT = 'B'

    def __init__(self, n_latents):
        super().__init__()
        self.image_encoder = CelebaImageEncoder(n_latents)
        self.image_decoder = CelebaImageDecoder(n_latents)
        self.attr_encoders = nn.ModuleList(
            [Celeba19AttrEncoder(n_latents) for _ in range(N_ATTRS)])
        self.attr_decoders = nn.ModuleList(
            [Celeba19AttrDecoder(n_latents) for _ in range(N_ATTRS)])
        self.n_latents = n_latents

    def infer(self, image=None, attrs=None, dropout_mask=None):
        attrs = attrs if attrs is not None else [None] * N_ATTRS
        mus, lvs = [], []
        if image is not None:
            self.image_encoder.classifier[2].mask = dropout_mask
            m, v = self.image_encoder(image)
            mus.append(m); lvs.append(v)
        for i in range(N_ATTRS):
            if attrs[i] is not None:
                m, v = self.attr_encoders[i](attrs[i].long())
                mus.append(m); lvs.append(v)
        return OF.poe_with_prior(mus, lvs, self.POE_VARIANT)

    def forward(self, image=None, attrs=None, eps=None, dropout_mask=None):
        mu, logvar = self.infer(image, attrs, dropout_mask)
        if self.training and eps is None:
            eps = torch.empty_like(mu).normal_()
        z = OF.reparametrize(mu, logvar, eps if self.training else None)
        image_recon = self.image_decoder(z)
        attr_recons = [self.attr_decoders[i](z).squeeze(1) for i in range(N_ATTRS)]
        return image_recon, attr_recons, mu, logvar, z


MODELS = {
    'mnist': (MnistMVAE, 64),
    'fashionmnist': (FmnistMVAE, 64),
    'celeba': (CelebaMVAE, 100),
    'celeba19': (Celeba19MVAE, 100),
}


def fill_parameters(model, seed):
    """Deterministic weight filler shared by the golden generator, the oracle and
    the HIP modules: parameters and BN buffers are visited
    in sorted ``state_dict`` key order and filled from ``Generator(seed)``.
    Scales keep activations O(1) through the stacks so the goldens exercise
    non-degenerate values."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if key.endswith('num_batches_tracked'):
                t.zero_()
            elif key.endswith('running_mean'):
                t.zero_()
            elif key.endswith('running_var'):
                t.fill_(1.0)
            elif t.dim() == 1:
                # biases and BN affine: BN weight ~ 1 +- 0.1, everything else ~ 0.1
                base = 1.0 if _is_bn_weight(model, key) else 0.0
                t.copy_(base + 0.1 * torch.randn(t.shape, generator=g))
            else:
                fan_in = t[0].numel() if t.dim() > 1 else t.numel()
                if _is_embedding(model, key):
                    fan_in = 1
                if _is_conv_transpose(model, key):
                    fan_in = t.shape[0] * 4  # Cin * (16 taps / stride^2 overlap), roughly
                t.copy_(torch.randn(t.shape, generator=g) * (1.0 / fan_in ** 0.5))
    return model


def _owner(model, key):
    mod = model
    for part in key.split('.')[:-1]:
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    return mod


def _is_bn_weight(model, key):
    return key.endswith('weight') and type(_owner(model, key)).__name__.startswith('BatchNorm')


def _is_embedding(model, key):
    return type(_owner(model, key)).__name__ == 'Embedding'


def _is_conv_transpose(model, key):
    return type(_owner(model, key)).__name__ == 'ConvTranspose2d'

"""Shared machinery of the four MVAE modules (mnist / fashionmnist / celeba / celeba19).

Mirrors the reference's ``MVAE`` surface (mnist/model.py:14-64):
``MVAE(n_latents)``, ``forward(image=None, text=None)`` (kwarg ``attrs`` for CelebA) returning
``(image_recon, label_recon, mu, logvar)``, ``infer``, ``reparametrize``, and the sub-module
attributes ``image_encoder / image_decoder / text_encoder / ...`` that ``sample.py`` calls
directly.  All arithmetic is HIP (``layers.py`` stacks + ``functional.py`` latent path).
"""
import torch
import torch.nn as nn

from . import kernels as K
from . import layers as L
from .arena import ParamArena
from .functional import PoEFn, PoEStackFn, ReparamFn


class Stack(nn.Module):
    """An encoder or decoder: ``self.stack_modules()`` lists its layers in execution order;
    the compiled plan runs them as fused HIP launches."""
    def stack_modules(self):
        raise NotImplementedError

    def plan(self):
        p = self.__dict__.get('_plan')
        if p is None:
            p = L.compile_plan(self.stack_modules())
            self.__dict__['_plan'] = p
        return p

    def run(self, x, groups=1, masks=None, bn_updates=1):
        # a stack called on its own (``model.image_decoder(z)``, mnist/sample.py:111; ``model.image_encoder(x)`` in
        # user code) before any model-level call: make sure the owner's parameters sit in the arena
        owner = self.__dict__.get('_owner_ref')
        owner = owner() if owner is not None else None
        if owner is not None and x.is_cuda and id(self) in owner.__dict__.get('_stack_ids', ()):
            owner.finalize()
        return L.run_plan(self.plan(), x, groups=groups, masks=masks, bn_updates=bn_updates,
                          training=self.training)


class ProductOfExperts(nn.Module):
    """The reference's module of the same name (mnist/model.py:149-163; celeba/model.py:193-207): parameters of
    the product of independent Gaussian experts, called on STACKED experts -- ``mu``, ``logvar`` of shape
    [M, B, D] (or [M, D]), row 0 being the prior expert in the reference's ``infer`` -- and returning
    ``(pd_mu, pd_logvar)`` of shape [B, D].  One HIP launch (``mvae_poe_fwd`` with MVAE_POE_NO_PRIOR), with a
    backward, so code written against ``model.experts(mu, logvar)`` keeps working.  ``MVAE.forward`` / ``infer``
    themselves never build the stack (the prior is a constant inside the fused launch).
    ``VARIANT``: 'A' = eps added twice and inside the log (mnist, fashionmnist), 'B' = once (celeba, celeba19)."""
    VARIANT = 'A'

    def forward(self, mu, logvar, eps=1e-8):
        if eps != 1e-8:
            raise ValueError('ProductOfExperts: eps is fixed at the reference default 1e-8 (got %r)' % (eps,))
        if mu.shape != logvar.shape or mu.dim() not in (2, 3):
            raise ValueError('ProductOfExperts: mu and logvar must both be [M, B, D] (or [M, D]); got %s and %s'
                             % (tuple(mu.shape), tuple(logvar.shape)))
        if not (mu.is_cuda and logvar.is_cuda):
            raise RuntimeError('multimodal-vae-public_amd: ProductOfExperts runs on the GPU only; there is no CPU '
                               'fallback')
        flat = mu.dim() == 2
        if flat:
            mu, logvar = mu.unsqueeze(1), logvar.unsqueeze(1)
        pd_mu, pd_logvar = PoEStackFn.apply(mu.float(), logvar.float(), self.VARIANT)
        return (pd_mu[0], pd_logvar[0]) if flat else (pd_mu, pd_logvar)


class ProductOfExpertsB(ProductOfExperts):
    VARIANT = 'B'


def prior_expert(size, use_cuda=False):
    """Universal prior expert N(0, 1): ``(mu, logvar)`` of zeros of shape ``size`` (mnist/model.py:172-185;
    celeba/model.py:216-229 writes log(ones) -- the same zeros)."""
    device = 'cuda' if use_cuda else 'cpu'
    return (torch.zeros(size, dtype=torch.float32, device=device),
            torch.zeros(size, dtype=torch.float32, device=device))


class MVAEBase(nn.Module):
    POE_VARIANT = 'A'

    def __init__(self, n_latents):
        super().__init__()
        self.n_latents = n_latents
        # the reference's ``self.experts = ProductOfExperts()`` (mnist/model.py:26): parameter-free, callable on a
        # stacked [M, B, D] pair; forward() / infer() use the fused launch instead
        self.experts = ProductOfExperts() if self.POE_VARIANT == 'A' else ProductOfExpertsB()
        self.__dict__['_arena'] = None
        self.__dict__['_rng'] = None

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        import weakref
        mods = [value] if isinstance(value, Stack) else (list(value) if isinstance(value, nn.ModuleList) else [])
        for m in mods:
            if isinstance(m, Stack):
                m.__dict__['_owner_ref'] = weakref.ref(self)      # not a child link: no cycle in modules()
                self.__dict__.setdefault('_stack_ids', set()).add(id(m))

    def _link_stacks(self):
        """(Re-)attach every stack to THIS model (a deepcopy keeps the original's back-references)."""
        import weakref
        ids = set()
        for m in self.modules():
            if isinstance(m, Stack):
                m.__dict__['_owner_ref'] = weakref.ref(self)
                ids.add(id(m))
        self.__dict__['_stack_ids'] = ids

    # ------------------------------------------------------------------ arena / device plumbing
    def arena_order(self):
        """Sub-modules in backward-completion order (decoders first)."""
        raise NotImplementedError

    def arena_adjacent(self):
        return ()

    def arena_tail(self):
        """Modules whose gradients complete LAST in the backward (the image encoder's first layers): laid
        out at the end of the arena so data-parallel replicas can all-reduce them as a small final bucket
        while Adam already runs on the earlier ones.  Default: none."""
        return ()

    def finalize(self):
        """Move parameters into the flat arena (idempotent; needs the model on the GPU)."""
        if self.__dict__.get('_arena') is None:
            p = next(self.parameters())
            if not p.is_cuda:
                raise RuntimeError('multimodal-vae-public_amd: move the model to the GPU first (model.cuda()); '
                                   'the HIP path has no CPU fallback')
            self.__dict__['_arena'] = ParamArena(self, order=self.arena_order(),
                                                 adjacent=self.arena_adjacent(), tail=self.arena_tail())
            self._link_stacks()
        return self.__dict__['_arena']

    @property
    def arena(self):
        return self.finalize()

    def _apply(self, fn, *a, **kw):
        # .cuda()/.cpu()/.to() re-materialise every parameter: the arena views are gone
        self.__dict__['_arena'] = None
        return super()._apply(fn, *a, **kw)

    def flush_counters(self):
        for m in self.modules():
            if isinstance(m, L._BatchNormMixin):
                m.flush_counters()

    # ------------------------------------------------------------------ noise
    def _noise_state(self, device):
        st = self.__dict__.get('_rng')
        if st is None or st[1].device != device:
            st = (0x5DEECE66D, torch.zeros(1, dtype=torch.int64, device=device))
            self.__dict__['_rng'] = st
        return st

    def seed_noise(self, seed):
        dev = next(self.parameters()).device
        self.__dict__['_rng'] = (int(seed), torch.zeros(1, dtype=torch.int64, device=dev))

    def device_randn(self, *shape):
        dev = next(self.parameters()).device
        seed, ctr = self._noise_state(dev)
        out = torch.empty(*shape, dtype=torch.float32, device=dev)
        K.randn_(out, seed, ctr)
        return out

    def device_bernoulli(self, keep, *shape):
        dev = next(self.parameters()).device
        seed, ctr = self._noise_state(dev)
        out = torch.empty(*shape, dtype=torch.float32, device=dev)
        K.bernoulli_(out, keep, seed ^ 0x9E3779B97F4A7C15, ctr)
        return out

    # ------------------------------------------------------------------ reference surface
    def reparametrize(self, mu, logvar, eps=None):
        """mnist/model.py:29-35: train -> eps * exp(0.5 logvar) + mu, eval -> mu."""
        if not self.training:
            return mu
        if eps is None:
            eps = self.device_randn(*mu.shape)
        return ReparamFn.apply(mu, logvar, eps)

    def _fuse(self, heads, eps, want_z):
        """PoE over the prior and the given encoder outputs ([B, 2D] each)."""
        self.finalize()
        B = heads[0].shape[0]
        dev = heads[0].device
        masks = K.all_experts_mask(len(heads), dev)
        noise = None
        if want_z and self.training:
            noise = eps if eps is not None else self.device_randn(1, B, self.n_latents)
            noise = noise.reshape(1, B, self.n_latents).contiguous()
        mu, lv, z, _ = PoEFn.apply((masks, noise, self.POE_VARIANT, self.n_latents), *heads)
        return mu[0], lv[0], z[0]

"""The recurrent text stacks of the MultiMNIST MVAE on HIP -- drop-in for the ``TextEncoder`` / ``TextDecoder``
classes of the reference's ``multimnist/model.py`` ("then MultiMNIST GRU stacks"): same
constructor arguments, ``forward`` signatures, return values and ``state_dict`` keys
(``embed.weight``, ``gru.weight_ih_l0`` ... ``gru.bias_hh_l0_reverse``, ``h2p.*`` / ``z2h.*``, ``h2o.*``).

    TextEncoder      multimnist/model.py:145-179   q(z|y): Embedding -> bidirectional GRU -> last position,
                                                   directions summed -> Linear(200, 2D) -> (mu, logvar)
    TextDecoder      multimnist/model.py:182-228   p(y|z): 4 greedy autoregressive steps of a 2-layer GRU
    swish / Swish    multimnist/model.py:247-253
    ProductOfExperts multimnist/model.py:231-244 (the single-eps variant, as celeba's), prior_expert :256-270
    max_length, n_characters, SOS, FILL            multimnist/utils.py:12-19

Every matrix product is an ``mvae_linear_*`` launch (leading dimensions make the reference's ``torch.cat((c_in, z))``
/ ``torch.cat((c_out, z))`` column ranges of one buffer), the gate arithmetic / embeddings / arg-max feedback are the
K16 kernels of csrc/gru.hip; forward and backward are hand-written (``torch.autograd.Function`` shells), no ATen
arithmetic.  What the reference evaluates but never uses is not evaluated: the backward direction of the encoder's
GRU contributes only its FIRST step (on the last character) to ``x[-1]`` (:173).

The rest of ``multimnist/model.py`` -- the 50x50 image stacks (a 5x5 and two pad-0 stride-2 convolutions) and the
``MVAE`` that joins them -- is outside SURVEY.md section 8 (section 2: ``multimnist/`` is not on the hot path)."""
import warnings

import torch
import torch.nn as nn

from .. import kernels as K
from ..base import ProductOfExpertsB as ProductOfExperts, prior_expert  # noqa: F401
from ..layers import Swish  # noqa: F401

max_length = 4          # multimnist/utils.py:12
n_characters = 12       # 10 digits + SOS + FILL (multimnist/utils.py:13-19)
SOS, FILL = 10, 11
KEEP = 0.9              # nn.GRU(..., dropout=0.1) between the decoder's two layers


def swish(x):
    return Swish()(x)


def _new(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError('multimodal-vae-public_amd: %s must live on the GPU (got %s); there is no CPU fallback'
                           % (what, t.device))


class GRU(nn.GRU):
    """Parameter holder with nn.GRU's names and default initialisation; the enclosing module runs the cells."""
    def forward(self, *a, **kw):
        raise RuntimeError('this GRU runs fused inside TextEncoder / TextDecoder; call the enclosing module')


def _cell_params(gru, layer, reverse=False):
    sfx = '_l%d%s' % (layer, '_reverse' if reverse else '')
    return tuple(getattr(gru, n + sfx) for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))


def _cell_fwd(x, h_prev, p, h_out):
    """One GRU cell: returns the tape entry (x, h_prev, gates)."""
    w_ih, w_hh, b_ih, b_hh = p
    B, H = h_prev.shape
    gi, gh = _new(B, 3 * H, like=x), _new(B, 3 * H, like=x)
    K.linear_fwd(x, w_ih, b_ih, gi, None)
    K.linear_fwd(h_prev, w_hh, b_hh, gh, None)
    gates = _new(B, 4 * H, like=x)
    K.gru_cell_fwd(gi, gh, h_prev, h_out, gates)
    return (x, h_prev, gates)


def _cell_bwd(dh, dh_extra, tape, p, grads, first, dx_out=None, dx_accumulate=False, want_dh_prev=True):
    """Backward of one cell.  ``grads`` = (dw_ih, dw_hh, db_ih, db_hh) (overwritten when ``first``, else added to).
    Returns (dx or None, dh_prev or None)."""
    x, h_prev, gates = tape
    w_ih, w_hh, _, _ = p
    B, H = h_prev.shape
    dgi, dgh, dh_prev = _new(B, 3 * H, like=x), _new(B, 3 * H, like=x), _new(B, H, like=x)
    K.gru_cell_bwd(dh, dh_extra, gates, h_prev, dgi, dgh, dh_prev)
    K.linear_wgrad(dgi, x, grads[0], grads[2], accumulate=not first)
    K.linear_wgrad(dgh, h_prev, grads[1], grads[3], accumulate=not first)
    dx = None
    if dx_out is not None:
        K.linear_dgrad(dgi, w_ih, dx_out, accumulate=dx_accumulate)
        dx = dx_out
    if want_dh_prev:
        K.linear_dgrad(dgh, w_hh, dh_prev, accumulate=True)
    return dx, (dh_prev if want_dh_prev else None)


# ----------------------------------------------------------------------------- encoder
class _TextEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bidirectional, w_emb, w_h2p, b_h2p, *gru_params):
        B, L = x.shape
        H = w_emb.shape[1]
        pf = gru_params[:4]
        pr = gru_params[4:8] if bidirectional else None
        e = _new(L, B, H, like=w_emb)
        for t in range(L):
            K.embedding_fwd(x[:, t], w_emb, e[t])
        zero = torch.zeros(B, H, dtype=torch.float32, device=w_emb.device)
        tape, h_prev = [], zero
        for t in range(L):
            h = _new(B, H, like=w_emb)
            tape.append(_cell_fwd(e[t], h_prev, pf, h))
            h_prev = h
        s = _new(B, H, like=w_emb)
        K.copy2d(h_prev, s)
        tape_r = None
        if bidirectional:
            hb = _new(B, H, like=w_emb)
            tape_r = _cell_fwd(e[L - 1], zero, pr, hb)     # the backward direction's state AT the last position
            K.copy2d(hb, s, accumulate=True)
        out = _new(B, w_h2p.shape[0], like=w_emb)
        K.linear_fwd(s, w_h2p, b_h2p, out, None)
        ctx.tapes = (x, e, tape, tape_r, s)
        ctx.params = (w_emb, w_h2p, pf, pr)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, e, tape, tape_r, s = ctx.tapes
        w_emb, w_h2p, pf, pr = ctx.params
        B, L = x.shape
        H = w_emb.shape[1]
        dout = dout.contiguous()
        dw_h2p, db_h2p = torch.empty_like(w_h2p), _new(w_h2p.shape[0], like=w_emb)
        K.linear_wgrad(dout, s, dw_h2p, db_h2p)
        ds = _new(B, H, like=w_emb)
        K.linear_dgrad(dout, w_h2p, ds)
        de = _new(L, B, H, like=w_emb)
        gf = tuple(torch.empty_like(p) for p in pf)
        dh = ds
        for t in range(L - 1, -1, -1):
            _, dh = _cell_bwd(dh, None, tape[t], pf, gf, first=(t == L - 1), dx_out=de[t])
        grads_r = ()
        if pr is not None:
            gr = tuple(torch.empty_like(p) for p in pr)
            _cell_bwd(ds, None, tape_r, pr, gr, first=True, dx_out=de[L - 1], dx_accumulate=True, want_dh_prev=False)
            grads_r = gr
        dw_emb = torch.empty_like(w_emb)
        for t in range(L):
            K.embedding_bwd(x[:, t], w_emb, de[t], dw_emb, accumulate=(t > 0))
        ctx.tapes = None
        return (None, None, dw_emb, dw_h2p, db_h2p) + gf + grads_r


class TextEncoder(nn.Module):
    """Parametrizes q(z|y) (multimnist/model.py:145-179)."""
    def __init__(self, n_latents, n_characters, n_hiddens=200, bidirectional=True):
        super().__init__()
        self.embed = nn.Embedding(n_characters, n_hiddens)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')     # "dropout expects num_layers > 1": the reference asks for the same GRU
            self.gru = GRU(n_hiddens, n_hiddens, 1, dropout=0.1, bidirectional=bidirectional)
        self.h2p = nn.Linear(n_hiddens, n_latents * 2)
        self.n_latents = n_latents
        self.n_hiddens = n_hiddens
        self.bidirectional = bidirectional

    def forward(self, x):
        _need_gpu(x, 'text'); _need_gpu(self.embed.weight, 'the module')
        if x.dim() != 2 or x.dtype != torch.int64:
            raise ValueError('text must be an int64 [batch, length] tensor of character indices')
        params = _cell_params(self.gru, 0) + (_cell_params(self.gru, 0, True) if self.bidirectional else ())
        p = _TextEncoderFn.apply(x.contiguous(), self.bidirectional, self.embed.weight, self.h2p.weight, self.h2p.bias,
                                 *params)
        return p[:, :self.n_latents], p[:, self.n_latents:]


# ----------------------------------------------------------------------------- decoder
class _TextDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, masks, holder, w_emb, w_z2h, b_z2h, w_h2o, b_h2o, *gru_params):
        z = z.contiguous()
        B, D = z.shape
        H = w_emb.shape[1]
        p0, p1 = gru_params[:4], gru_params[4:8]
        n_chars = w_h2o.shape[0]
        dev = z.device
        hz = _new(B, H, like=z)
        K.linear_fwd(z, w_z2h, b_z2h, hz, None)
        h0_prev = h1_prev = hz
        c_in = torch.full((B,), SOS, dtype=torch.int64, device=dev)
        words = _new(B, max_length, n_chars, like=z)
        steps = []
        for i in range(max_length):
            xcat = _new(B, H + D, like=z)
            K.embedding_fwd(c_in, w_emb, xcat[:, :H], swish=True)          # swish(self.embed(c_in))  (:220)
            K.copy2d(z, xcat[:, H:])                                       # torch.cat((c_in, z), dim=1) (:221)
            h0 = _new(B, H, like=z)
            t0 = _cell_fwd(xcat, h0_prev, p0, h0)
            if masks is not None:                                          # nn.GRU's Dropout between its two layers
                d0 = _new(B, H, like=z)
                K.copy2d(h0, d0, mask=masks[i], scale=1.0 / KEEP)
            else:
                d0 = h0
            ocat = _new(B, H + D, like=z)
            t1 = _cell_fwd(d0, h1_prev, p1, ocat[:, :H])                   # c_out lands in the cat buffer (:224-225)
            K.copy2d(z, ocat[:, H:])
            K.linear_fwd(ocat, w_h2o, b_h2o, words[:, i, :], None)         # words[:, i] = self.h2o(...) (:212,226)
            steps.append((c_in, t0, t1, ocat))
            nxt = torch.empty(B, dtype=torch.int64, device=dev)
            K.argmax_rows(words[:, i, :], nxt)                             # greedy feedback (:211,213)
            c_in = nxt
            h0_prev, h1_prev = h0, ocat[:, :H]
        holder['fed'] = torch.stack([s[0] for s in steps])
        ctx.tapes = (z, masks, hz, steps)
        ctx.params = (w_emb, w_z2h, w_h2o, p0, p1)
        return words

    @staticmethod
    def backward(ctx, dwords):
        z, masks, hz, steps = ctx.tapes
        w_emb, w_z2h, w_h2o, p0, p1 = ctx.params
        B, D = z.shape
        H = w_emb.shape[1]
        dwords = dwords.contiguous()
        dz = torch.zeros_like(z)
        dw_h2o, db_h2o = torch.empty_like(w_h2o), _new(w_h2o.shape[0], like=z)
        g0 = tuple(torch.empty_like(p) for p in p0)
        g1 = tuple(torch.empty_like(p) for p in p1)
        dw_emb = torch.empty_like(w_emb)
        dh0_carry = torch.zeros(B, H, dtype=torch.float32, device=z.device)
        dh1_carry = torch.zeros(B, H, dtype=torch.float32, device=z.device)
        for i in range(max_length - 1, -1, -1):
            c_in, t0, t1, ocat = steps[i]
            first = i == max_length - 1
            dlog = dwords[:, i, :]
            K.linear_wgrad(dlog, ocat, dw_h2o, db_h2o, accumulate=not first)
            d_ocat = _new(B, H + D, like=z)
            K.linear_dgrad(dlog, w_h2o, d_ocat)
            K.copy2d(d_ocat[:, H:], dz, accumulate=True)
            dd0 = _new(B, H, like=z)
            _, dh1_carry = _cell_bwd(d_ocat[:, :H], dh1_carry, t1, p1, g1, first, dx_out=dd0)
            if masks is not None:
                dd0m = _new(B, H, like=z)
                K.copy2d(dd0, dd0m, mask=masks[i], scale=1.0 / KEEP)
                dd0 = dd0m
            dxcat = _new(B, H + D, like=z)
            _, dh0_carry = _cell_bwd(dd0, dh0_carry, t0, p0, g0, first, dx_out=dxcat)
            K.copy2d(dxcat[:, H:], dz, accumulate=True)
            K.embedding_bwd(c_in, w_emb, dxcat[:, :H], dw_emb, swish=True, accumulate=not first)
        dhz = _new(B, H, like=z)
        K.copy2d(dh0_carry, dhz)
        K.copy2d(dh1_carry, dhz, accumulate=True)          # z2h(z) initialises BOTH layers (.repeat(2, 1, 1), :207)
        dw_z2h, db_z2h = torch.empty_like(w_z2h), _new(w_z2h.shape[0], like=z)
        K.linear_wgrad(dhz, z, dw_z2h, db_z2h)
        K.linear_dgrad(dhz, w_z2h, dz, accumulate=True)
        ctx.tapes = None
        return (dz, None, None, dw_emb, dw_z2h, db_z2h, dw_h2o, db_h2o) + g0 + g1


class TextDecoder(nn.Module):
    """Parametrizes p(y|z) (multimnist/model.py:182-228).  ``forward(z)`` returns the [batch, 4, n_characters] logits;
    ``dropout_masks`` (4 tensors [batch, 200] in {0, 1}) replays a host draw in parity runs, otherwise the training-mode
    masks come from the device Philox stream.  ``last_fed`` holds the characters fed back ([4, batch])."""
    def __init__(self, n_latents, n_characters, n_hiddens=200):
        super().__init__()
        self.embed = nn.Embedding(n_characters, n_hiddens)
        self.z2h = nn.Linear(n_latents, n_hiddens)
        self.gru = GRU(n_hiddens + n_latents, n_hiddens, 2, dropout=0.1)
        self.h2o = nn.Linear(n_hiddens + n_latents, n_characters)
        self.n_latents = n_latents
        self.n_characters = n_characters
        self.n_hiddens = n_hiddens
        self.last_fed = None
        self.__dict__['_rng'] = None

    def seed_noise(self, seed):
        self.__dict__['_rng'] = (int(seed), torch.zeros(1, dtype=torch.int64, device=self.embed.weight.device))

    def _device_masks(self, B):
        st = self.__dict__.get('_rng')
        if st is None or st[1].device != self.embed.weight.device:
            self.seed_noise(0x5DEECE66D)
            st = self.__dict__['_rng']
        masks = torch.empty(max_length, B, self.n_hiddens, dtype=torch.float32, device=self.embed.weight.device)
        K.bernoulli_(masks, KEEP, st[0], st[1])
        return [masks[i] for i in range(max_length)]

    def forward(self, z, dropout_masks=None):
        _need_gpu(z, 'z'); _need_gpu(self.embed.weight, 'the module')
        masks = None
        if self.training:
            if dropout_masks is None:
                masks = self._device_masks(z.shape[0])
            else:
                masks = [m.to(z.device).float().contiguous() for m in dropout_masks]
                if len(masks) != max_length or any(m.shape != (z.shape[0], self.n_hiddens) for m in masks):
                    raise ValueError('dropout_masks: %d tensors of [batch, %d]' % (max_length, self.n_hiddens))
        holder = {}
        words = _TextDecoderFn.apply(z.float(), masks, holder, self.embed.weight, self.z2h.weight, self.z2h.bias,
                                     self.h2o.weight, self.h2o.bias, *(_cell_params(self.gru, 0) + _cell_params(self.gru, 1)))
        self.last_fed = holder.get('fed')
        return words

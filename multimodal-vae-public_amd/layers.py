"""HIP-backed layers and the stack executor.

The layer classes subclass the ``torch.nn`` containers the reference uses (so parameter names,
shapes, default initialisation and ``state_dict`` keys are identical -- SURVEY.md Appendix A),
but their arithmetic never touches ATen: a *stack* (an encoder or a decoder: a list of these
layers) is compiled into a short plan of fused HIP launches

    Linear(+Swish)(+Dropout) | Conv2d/ConvTranspose2d 4x4 (+Swish) | BatchNorm(+Swish) |
    Embedding+Swish | view

and run by ``forward_tape`` / ``backward_tape``.  The backward is hand-written: each dgrad
launch multiplies by swish'(pre-activation) of the layer that produced its input in the GEMM
epilogue, weight gradients are written straight into the gradient arena, and BatchNorm handles
its own Swish.  ``StackFn`` exposes the pair to autograd for the reference's module surface
(``model(image, text)`` ... ``loss.backward()``); the fused train step in ``engine.py`` calls
the tape functions directly.

A stack may process ``G`` groups of ``B`` rows at once (``groups=G``): BatchNorm statistics are
per group, so the G separate ``model()`` calls of the reference's train step become one launch
per layer with identical results.
"""
import os

import torch
import torch.nn as nn

from . import kernels as K
from .arena import grad_target


# ----------------------------------------------------------------------------- layer containers
class Swish(nn.Module):
    """x * sigmoid(x) (reference: mnist/model.py:166-169); fused into its producer when it
    follows a Linear / conv / BatchNorm inside a stack."""
    def forward(self, x):
        return _SwishFn.apply(x.contiguous())


class _SwishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        K.swish_fwd(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.swish_bwd(g.contiguous(), x, dx)
        return dx


class Linear(nn.Linear):
    def forward(self, x):
        return run_stack([self], x)


class _RepackMixin(object):
    """Conv modules may carry a repacked weight copy (``_probe_repack``); moving the module drops it."""
    def _apply(self, fn, *a, **kw):
        for name in ('_wr', '_wr_item', '_wr_fresh', '_wr_need'):
            self.__dict__.pop(name, None)
        return super()._apply(fn, *a, **kw)


class Conv2d(_RepackMixin, nn.Conv2d):
    """4x4, bias=False, (stride, pad) in {(2,1), (1,0)} -- the only shapes on the hot path."""
    def forward(self, x):
        return run_stack([self], x)


class ConvTranspose2d(_RepackMixin, nn.ConvTranspose2d):
    def forward(self, x):
        return run_stack([self], x)


class _BatchNormMixin(object):
    def _init_pending(self):
        self._nbt_pending = 0

    def flush_counters(self):
        """num_batches_tracked is advanced on the host and written back lazily: a device-side
        int64 add per BatchNorm per call would be 11-21 stray launches per step."""
        if getattr(self, '_nbt_pending', 0):
            self.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counters()
        return super()._save_to_state_dict(destination, prefix, keep_vars)

    def forward(self, x):
        return run_stack([self], x)


class BatchNorm2d(_BatchNormMixin, nn.BatchNorm2d):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._init_pending()


class BatchNorm1d(_BatchNormMixin, nn.BatchNorm1d):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._init_pending()


class Embedding(nn.Embedding):
    """Only ever followed by Swish in the reference; the stack fuses the pair."""
    def forward(self, idx):
        raise RuntimeError('Embedding runs fused with the Swish that follows it; call the '
                           'enclosing encoder instead')


class Dropout(nn.Dropout):
    """p = 0.1 after the CelebA encoder's Linear+Swish (celeba/model.py:91); the keep-mask is
    an explicit input of the stack (host-drawn for parity, Philox on device otherwise)."""
    def forward(self, x):
        raise RuntimeError('Dropout runs fused into the preceding Linear; call the enclosing encoder')


class View(nn.Module):
    """x.view(-1, *shape) between the conv and linear halves of a stack (parameter-free and
    not registered in any Sequential, so state_dict keys are unchanged)."""
    def __init__(self, *shape):
        super().__init__()
        self.shape = tuple(shape)


# ----------------------------------------------------------------------------- plan
class _Op(object):
    __slots__ = ('kind', 'mod', 'act', 'drop', 'pair')

    def __init__(self, kind, mod, act=False, drop=0.0, pair=None):
        self.kind, self.mod, self.act, self.drop, self.pair = kind, mod, act, drop, pair


class HeadPair(nn.Module):
    """Two Linear heads on the same input (fc31 / fc32 of the MNIST encoders,
    mnist/model.py:77-78,84) executed as ONE GEMM of width 2D: the arena lays the two weight
    matrices (and biases) back to back.  Parameter-free wrapper; the heads stay registered on
    their owner under the reference's names."""
    def __init__(self, head_a, head_b):
        super().__init__()
        object.__setattr__(self, 'heads', (head_a, head_b))   # not registered as children


def flatten_modules(mods):
    out = []
    for m in mods:
        if isinstance(m, nn.Sequential):
            out.extend(flatten_modules(list(m)))
        else:
            out.append(m)
    return out


def compile_plan(mods):
    mods = flatten_modules(mods)
    plan, i = [], 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, (nn.Linear, HeadPair)):
            kind = 'lin' if isinstance(m, nn.Linear) else 'lin2'
            act = isinstance(nxt, Swish)
            i += 2 if act else 1
            drop = 0.0
            if act and i < len(mods) and isinstance(mods[i], nn.Dropout):
                drop = mods[i].p
                i += 1
            plan.append(_Op(kind, m, act=act, drop=drop))
        elif isinstance(m, (nn.ConvTranspose2d, nn.Conv2d)):
            _check_conv(m)
            kind = 'convT' if isinstance(m, nn.ConvTranspose2d) else 'conv'
            act = isinstance(nxt, Swish)
            plan.append(_Op(kind, m, act=act))
            i += 2 if act else 1
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            act = isinstance(nxt, Swish)
            plan.append(_Op('bn', m, act=act))
            i += 2 if act else 1
        elif isinstance(m, nn.Embedding):
            if not isinstance(nxt, Swish):
                raise RuntimeError('Embedding must be followed by Swish')
            plan.append(_Op('emb', m, act=True))
            i += 2
        elif isinstance(m, View):
            plan.append(_Op('view', m))
            i += 1
        elif isinstance(m, Swish):
            raise RuntimeError('dangling Swish: no fusable producer in front of it')
        else:
            raise RuntimeError('layer %s is not on the MVAE hot path' % type(m).__name__)
    for a, b in zip(plan[:-1], plan[1:]):
        if b.kind == 'bn' and a.act:
            raise RuntimeError('BatchNorm directly after an activated layer is not supported')
    return plan


def _check_conv(m):
    ok = (tuple(m.kernel_size) == (4, 4) and m.bias is None and
          (tuple(m.stride), tuple(m.padding)) in (((2, 2), (1, 1)), ((1, 1), (0, 0))))
    if not ok:
        raise RuntimeError('only 4x4 convs with bias=False and (stride,pad) in {(2,1),(1,0)} are built')


def plan_params(plan):
    ps = []
    for op in plan:
        if op.kind == 'lin2':
            for h in op.mod.heads:
                ps.extend(h.parameters())
        elif op.kind != 'view':
            ps.extend(op.mod.parameters())
    return ps


def tail_cut(plan, tail_modules):
    """Index k such that plan[:k] holds exactly the ops of ``tail_modules`` (the first layers of a stack).
    ``backward_tape(plan[k:], tape[k:], g, need_input_grad=True)`` followed by
    ``backward_tape(plan[:k], tape[:k], that gradient)`` equals one pass over the whole plan: the upper
    slice's first Linear / conv finds no producer and leaves the Swish' of plan[k-1] to the lower slice,
    which applies it first (its last op is activated) -- one extra elementwise launch."""
    ids = set(id(x) for mod in tail_modules for x in flatten_modules([mod]))
    k = 0
    for i, op in enumerate(plan):
        if id(op.mod) in ids:
            k = i + 1
    if k == 0 or any(id(op.mod) not in ids and op.kind != 'view' for op in plan[:k]):
        raise RuntimeError('the arena tail is not a prefix of the stack')
    while k < len(plan) and plan[k].kind == 'view':      # a view between the slices belongs to the lower one
        k += 1
    return k


def n_dropout(plan):
    return sum(1 for op in plan if op.drop > 0)


# ----------------------------------------------------------------------------- executor
# statistics-only decoder passes: the conv in front of the last BatchNorm leaves statistics records instead of its
# output (MVAE_STATS_CONV=0: the storing launch + the statistics sweep over what it stored)
STATS_CONV = os.environ.get('MVAE_STATS_CONV', '1') != '0'


def _lin_weights(op):
    """(weight [N,K], bias [N] or None) -- for a HeadPair the joined arena views."""
    if op.kind == 'lin':
        return op.mod.weight, op.mod.bias
    a, b = op.mod.heads
    arena = getattr(a.weight, '_arena', None)
    if arena is None:
        raise RuntimeError('paired heads need the parameter arena (call model.finalize())')
    w, _ = arena.joined(a.weight, b.weight)
    bias, _ = arena.joined(a.bias, b.bias)
    return w, bias


def forward_tape(plan, x, groups=1, masks=None, bn_updates=1, training=True, final_out=None,
                 bn_updates_dev=None, stats_only=False, loss_fold=None):
    """Run the plan.  Returns (output, tape); tape is None when not training.
    ``final_out``: preallocated [rows, N] destination for a stack that ends in a plain Linear
    (celeba19 collects its 18 attribute decoders' logits in one buffer).  ``bn_updates_dev``:
    device int32[1] overriding ``bn_updates`` (number of running-statistics updates).
    ``stats_only``: the pass exists only for its BatchNorm running-statistics side effect -- stop at
    the last BatchNorm, which computes its statistics without writing an output; returns
    (None, None).  ``loss_fold(x, w, b, out)``: the stack ends in a plain Linear whose only consumer is a
    reconstruction term -- the callable launches Linear + term in one kernel (K.linear_bce_fwd / K.linear_ce_fwd)
    and fills ``out`` [rows, N] with d loss / d logits, which is then what this function returns in place of the
    logits (and what ``backward_tape`` takes as ``g``)."""
    masks = list(masks) if masks is not None else []
    if loss_fold is not None and not (plan[-1].kind == 'lin' and not plan[-1].act and training):
        raise RuntimeError('loss_fold needs a training-mode stack that ends in a plain Linear')
    tape = [] if training else None
    h = x
    last_op = plan[-1]
    last_bn = max([i for i, op in enumerate(plan) if op.kind == 'bn'] or [-1]) if stats_only else -1
    if stats_only and (last_bn < 0 or not training):
        raise RuntimeError('stats_only needs a training-mode stack with a BatchNorm')
    for op_index, op in enumerate(plan):
        saved = None
        if op.kind in ('lin', 'lin2'):
            if h.dim() != 2 or h.stride(1) != 1:
                raise RuntimeError('Linear expects a [rows, features] input with unit column stride')
            w, b = _lin_weights(op)
            M, N = h.shape[0], w.shape[0]
            mask = None
            if op.drop > 0 and training:
                if not masks:
                    raise RuntimeError('stack has a Dropout but no keep-mask was supplied')
                mask = masks.pop(0)
            if op.act:
                # the pre-activation is only kept for a backward pass (a statistics-only pass has none: 118 MB less to
                # write for celeba19's 4608-row Linear(100, 6400))
                pre = torch.empty(M, N, dtype=torch.float32, device=h.device) if (training and not stats_only) else None
                act = torch.empty(M, N, dtype=torch.float32, device=h.device)
                K.linear_fwd(h, w.detach(), None if b is None else b.detach(), pre, act, mask,
                             1.0 / (1.0 - op.drop) if mask is not None else 1.0)
                saved = (h, pre, mask)
                h = act
            else:
                if final_out is not None and op is last_op:
                    pre = final_out.reshape(M, N)
                else:
                    pre = torch.empty(M, N, dtype=torch.float32, device=h.device)
                if loss_fold is not None and op is last_op:
                    loss_fold(h, w.detach(), None if b is None else b.detach(), pre)
                else:
                    K.linear_fwd(h, w.detach(), None if b is None else b.detach(), pre, None)
                saved = (h, None, None)
                h = pre
        elif op.kind in ('conv', 'convT'):
            m = op.mod
            s, p = m.stride[0], m.padding[0]
            h = h.contiguous()
            Bn, _, H, W = h.shape
            if op.kind == 'conv':
                Cout, OH, OW = m.out_channels, (H + 2 * p - 4) // s + 1, (W + 2 * p - 4) // s + 1
            else:
                Cout, OH, OW = m.out_channels, (H - 1) * s - 2 * p + 4, (W - 1) * s - 2 * p + 4
            if op.kind == 'convT' and stats_only and STATS_CONV and op_index + 1 == last_bn and not op.act and training:
                tiles = K.convT2d_stats_tiles(h, m.weight, s, p)
                if tiles > 0 and Bn % groups == 0 and tiles % groups == 0:
                    # the pass ends at the BatchNorm behind this layer and only its statistics are wanted: the launch
                    # computes the layer, stores nothing and leaves per-tile (mean, M2) records; the merge launch turns
                    # them into the BatchNorm's running statistics (kernels.convT2d_fwd_stats / bn_stats_merge)
                    wr = _fresh_repack(m, _probe_repack(m, True, Bn, h.shape[1], H, W, Cout, s, p))
                    part = K.convT2d_fwd_stats(h, m.weight.detach(), s, p, wr=wr)
                    bn = plan[last_bn].mod
                    K.bn_stats_merge(part, groups, None, None, bn.running_mean, bn.running_var, eps=bn.eps,
                                     momentum=bn.momentum, n_updates=bn_updates, n_updates_dev=bn_updates_dev)
                    bn._nbt_pending += groups * bn_updates
                    return None, None
            pre = act = None
            if (not op.act) or (training and not stats_only):
                pre = torch.empty(Bn, Cout, OH, OW, dtype=torch.float32, device=h.device)
            if op.act:
                act = torch.empty(Bn, Cout, OH, OW, dtype=torch.float32, device=h.device)
            if op.kind == 'conv':
                K.conv2d_fwd(h, m.weight.detach(), pre, act, s, p)
            else:
                wr = None
                if training:        # eval / sample.py launches always repack for themselves
                    wr = _fresh_repack(m, _probe_repack(m, True, Bn, h.shape[1], H, W, Cout, s, p))
                K.convT2d_fwd(h, m.weight.detach(), pre, act, s, p, wr=wr)
            saved = (h, pre if op.act else None, None)
            h = act if op.act else pre
        elif op.kind == 'bn':
            m = op.mod
            h = h.contiguous()
            y = None if (stats_only and op_index == last_bn) else torch.empty_like(h)
            if training:
                C = h.shape[1]
                sm = torch.empty(groups, C, dtype=torch.float32, device=h.device)
                si = torch.empty(groups, C, dtype=torch.float32, device=h.device)
                K.bn_train_fwd(h, m.weight.detach(), m.bias.detach(), y, sm, si, m.running_mean,
                               m.running_var, groups, eps=m.eps, momentum=m.momentum,
                               n_updates=bn_updates, swish=op.act, n_updates_dev=bn_updates_dev)
                m._nbt_pending += groups * bn_updates
                if y is None:
                    return None, None
                saved = (h, sm, si)
            else:
                K.bn_eval_fwd(h, m.weight.detach(), m.bias.detach(), y, m.running_mean, m.running_var,
                              eps=m.eps, swish=op.act)
            h = y
        elif op.kind == 'emb':
            m = op.mod
            idx = h if (h.dim() == 1 and h.dtype == torch.float32) else h.contiguous()   # fp32 may be strided
            act = torch.empty(idx.numel(), m.embedding_dim, dtype=torch.float32, device=idx.device)
            K.embedding_swish_fwd(idx, m.weight.detach(), act)
            saved = (idx,)
            h = act
        elif op.kind == 'view':
            h = h.reshape((-1,) + op.mod.shape)
        if training:
            tape.append(saved)
    return h, tape


def _producer(plan, tape, i):
    """(pre, mask, drop) of the activated Linear/conv that produced op i's input, skipping views."""
    j = i - 1
    while j >= 0 and plan[j].kind == 'view':
        j -= 1
    if j >= 0 and plan[j].act and plan[j].kind in ('lin', 'lin2', 'conv', 'convT'):
        return tape[j][1], tape[j][2], plan[j].drop
    return None, None, 0.0


def backward_tape(plan, tape, g, need_input_grad=False, groups=1, input_grad_out=None,
                  input_grad_accumulate=False, defer_input_grad=False, deferred=None):
    """Backward through the plan.  ``g`` = gradient w.r.t. the stack output.  Parameter
    gradients go to ``p.grad`` (the arena); returns the input gradient or None.
    ``input_grad_out`` (stacks that start with a Linear): write -- or with
    ``input_grad_accumulate`` add -- the input gradient into this buffer (the shared dz of the
    decoders) instead of allocating one.  ``defer_input_grad`` (same stacks): skip the first
    Linear's input gradient and return the gradient w.r.t. its OUTPUT; the caller finishes with
    ``first_linear_dgrad`` -- used when two stacks run on different streams but share dz.
    ``deferred``: a list that receives the weight-gradient launches as closures instead of
    running them here, so only the data-gradient chain is serial; the caller runs them (in order)
    on another stream and keeps the list alive until that stream is joined."""
    def side(fn):
        if deferred is None:
            fn()
        else:
            deferred.append(fn)

    last = len(plan) - 1
    while last >= 0 and plan[last].kind == 'view':
        last -= 1
    if plan[last].act and plan[last].kind in ('lin', 'lin2', 'conv', 'convT'):
        pre = tape[last][1]
        g2 = torch.empty_like(pre)
        K.swish_bwd(g.reshape(pre.shape).contiguous(), pre, g2)
        if tape[last][2] is not None:
            raise RuntimeError('a stack may not end in Dropout')
        g = g2
    first = 0
    while first < len(plan) and plan[first].kind == 'view':
        first += 1
    for i in range(len(plan) - 1, -1, -1):
        op, saved = plan[i], tape[i]
        want_dx = need_input_grad or i > first
        if op.kind == 'view':
            continue
        pre_in, mask_in, drop_in = _producer(plan, tape, i)
        if op.kind in ('lin', 'lin2'):
            x = saved[0]
            g = g.reshape(x.shape[0], -1)
            if g.stride(1) != 1:
                g = g.contiguous()
            if isinstance(deferred, WgradBatch):
                deferred.add_linear(op, g, x)
            else:
                side(lambda op=op, g=g, x=x: _lin_wgrad(op, g, x))
            if defer_input_grad and i == first:
                return g
            if want_dx:
                w, _ = _lin_weights(op)
                acc = False
                if i == first and input_grad_out is not None:
                    dx, acc = input_grad_out, input_grad_accumulate
                else:
                    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                K.linear_dgrad(g, w.detach(), dx, pre_in, mask_in,
                               1.0 / (1.0 - drop_in) if mask_in is not None else 1.0, accumulate=acc)
                g = dx
        elif op.kind in ('conv', 'convT'):
            m = op.mod
            x = saved[0]
            s, p = m.stride[0], m.padding[0]
            out_shape = _conv_out_shape(op, x)
            g = g.reshape(out_shape).contiguous()
            def conv_wgrad(op=op, m=m, g=g, x=x, s=s, p=p):
                dw, acc = grad_target(m.weight)
                (K.conv2d_wgrad if op.kind == 'conv' else K.convT2d_wgrad)(g, x, dw, s, p, accumulate=acc)
            side(conv_wgrad)
            if want_dx:
                dx = torch.empty_like(x)
                pin = None if pre_in is None else pre_in.reshape(x.shape)
                if op.kind == 'conv':
                    key = _probe_repack(m, False, x.shape[0], x.shape[1], x.shape[2], x.shape[3], m.out_channels, s, p)
                    K.conv2d_dgrad(g, m.weight.detach(), dx, pin, s, p, wr=_fresh_repack(m, key))
                else:
                    K.convT2d_dgrad(g, m.weight.detach(), dx, pin, s, p)
                g = dx
        elif op.kind == 'bn':
            m = op.mod
            x, sm, si = saved
            g = g.reshape(x.shape).contiguous()
            dgam, acc1 = grad_target(m.weight)
            dbet, acc2 = grad_target(m.bias)
            if acc1 != acc2:
                raise RuntimeError('BatchNorm weight/bias gradients out of sync')
            dx = torch.empty_like(x)
            K.bn_train_bwd(g, x, m.weight.detach(), m.bias.detach(), sm, si, dx, dgam, dbet, groups,
                           swish=op.act, accumulate=acc1)
            if pre_in is not None:
                raise RuntimeError('BatchNorm input must come from a non-activated layer')
            g = dx
        elif op.kind == 'emb':
            def emb_wgrad(m=op.mod, idx=saved[0], g=g.contiguous()):
                dw, acc = grad_target(m.weight)
                K.embedding_swish_bwd(idx, m.weight.detach(), g, dw, accumulate=acc)
                if not acc and isinstance(deferred, WgradBatch):
                    deferred.final_gradient(dw)
            side(emb_wgrad)
            g = None
    return g if need_input_grad else None


def first_linear_dgrad(plan, g, out, accumulate):
    """The step ``backward_tape(..., defer_input_grad=True)`` left out: out (+)= g @ W_first."""
    first = 0
    while plan[first].kind == 'view':
        first += 1
    w, _ = _lin_weights(plan[first])
    K.linear_dgrad(g, w.detach(), out, None, None, 1.0, accumulate=accumulate)


def _conv_out_shape(op, x):
    m = op.mod
    s, p = m.stride[0], m.padding[0]
    Bn, _, H, W = x.shape
    if op.kind == 'conv':
        return (Bn, m.out_channels, (H + 2 * p - 4) // s + 1, (W + 2 * p - 4) // s + 1)
    return (Bn, m.out_channels, (H - 1) * s - 2 * p + 4, (W - 1) * s - 2 * p + 4)


# ---- repacked weight copies of the dgrad-form conv launches (Conv2d data gradient, ConvTranspose2d forward).
# The launches make the copy themselves unless handed one; a step engine makes all of them in ONE launch at the
# start of its step (``repack_weights``) and withdraws them at its end (``repack_done``) -- between steps the
# optimizer changes the weights, and anything run outside an engine step takes the self-contained path.
def _probe_repack(m, transposed, B, Cin, H, W, Cout, s, p):
    """Ask the library -- once per GEOMETRY -- whether this launch reads a repacked weight copy
    (``mvae_conv_k4_repack_floats``: the answer and the kernel selection depend on B, H, W; the direct kernels of
    the <= 4-channel layers read ``w`` itself).  The copy's content depends only on the layer, so a module keeps
    one buffer; which geometries may use it is remembered per (B, H, W) (one cached answer per module
    handed ``w = NULL`` to a geometry that needs ``w``)."""
    need = m.__dict__.setdefault('_wr_need', {})
    key = (B, H, W)
    if key not in need:
        n = K.conv_repack_floats(transposed, m.weight.detach(), B, Cin, H, W, Cout, s, p)
        need[key] = n > 0
        if n > 0 and getattr(m, '_wr', None) is None:
            m._wr = torch.empty(n, dtype=torch.float32, device=m.weight.device)
            m._wr_item = (bool(transposed), Cin, Cout, s, p)
            m._wr_fresh = False
    return key


def _fresh_repack(m, key):
    """The module's repacked copy if an engine step made it for THIS step and this geometry reads one."""
    if getattr(m, '_wr_fresh', False) and m.__dict__.get('_wr_need', {}).get(key, False):
        return m._wr
    return None


def repack_weights(mods):
    """One launch for the repacked copies of every probed conv module in ``mods``; marks them usable."""
    items = []
    for m in mods:
        if getattr(m, '_wr', None) is not None:
            tr, Cin, Cout, s, p = m._wr_item
            items.append((m.weight.detach(), m._wr, tr, Cin, Cout, s, p))
    if items:
        K.conv_repack_batched(items)
        for m in mods:
            if getattr(m, '_wr', None) is not None:
                m._wr_fresh = True


def repack_done(mods):
    for m in mods:
        if getattr(m, '_wr_fresh', False):
            m._wr_fresh = False


def _lin_wgrad_targets(op):
    """(dw, db, accumulate) of a Linear / paired-heads op; marks the gradients as written."""
    if op.kind == 'lin':
        m = op.mod
        dw, acc = grad_target(m.weight)
        db = None
        if m.bias is not None:
            db, acc_b = grad_target(m.bias)
            if acc_b != acc:
                raise RuntimeError('Linear weight/bias gradients out of sync')
        return dw, db, acc
    a, b = op.mod.heads
    arena = a.weight._arena
    acc = a.weight.grad is not None
    for p in (a.weight, a.bias, b.weight, b.bias):
        if (p.grad is not None) != acc:
            raise RuntimeError('paired-head gradients out of sync')
        grad_target(p)
    _, dw = arena.joined(a.weight, b.weight)
    _, db = arena.joined(a.bias, b.bias)
    return dw, db, acc


def _lin_wgrad_key(op):
    """Identity of the weight a Linear / paired-heads op writes a gradient for -- WITHOUT touching it (``grad_target`` sets
    ``p.grad`` and decides the accumulate flag: a side effect that must happen in list order, see WgradBatch.flush)."""
    return id(op.mod.weight) if op.kind == 'lin' else id(op.mod.heads[0].weight)


def _lin_wgrad(op, g, x):
    dw, db, acc = _lin_wgrad_targets(op)
    K.linear_wgrad(g, x, dw, db, accumulate=acc)


class WgradBatch(list):
    """A ``deferred`` list for ``backward_tape`` that turns the Linear weight gradients of the chain into ONE
    launch (``K.linear_wgrad_batched``): nothing on the data-gradient chain reads them, so they need not sit
    between its launches.  ``flush()`` issues the batch (and any other queued closure) on the current stream;
    the entries keep dy / x alive until then."""

    adam = None      # an ``optim.AdamFusion``: the batch launch also applies Adam to what it computed

    def __init__(self, adam=None):
        list.__init__(self)
        self.adam = adam
        self._final = []

    def add_linear(self, op, g, x):
        self.append(('lin', op, g, x))

    def final_gradient(self, grad):
        """A queued closure reports a gradient it has just written in full (first and only contribution of the
        step): with ``adam`` its update rides the batch launch too."""
        if self.adam is not None:
            self._final.append(grad)

    def flush(self):
        items, seen = [], set()
        adam = self.adam
        # side-effect-free pre-pass: which weights receive exactly ONE contribution in this list.  The
        # gradient targets themselves -- grad_target() sets p.grad and the accumulate flag -- are taken inside the ordered
        # loop below, interleaved with the queued closures as they always were.
        keys = [(_lin_wgrad_key(e[1]) if isinstance(e, tuple) else None) for e in self]
        key_of = {}                 # gradient address -> weight identity, filled as targets are taken
        if adam is not None:
            # A fused launch applies Adam to the gradient it has just written, so that gradient must be FINAL: exactly
            # one contribution in this list, not accumulated onto an earlier one, and nothing behind it (a
            # layer used twice per step got its update on a partial gradient and rest() then skipped it).  Gradients
            # that do not qualify leave this flush un-fused and stay with the arena-wide launch (rest()).
            count = {}
            for k in keys:
                if k is not None:
                    count[k] = count.get(k, 0) + 1
            fusable = {k for k, n in count.items() if n == 1}
        else:
            fusable = set()

        def guard(dw, db):
            # a contribution to a gradient some EARLIER fused launch of this step has already consumed cannot be repaired
            if self.adam is not None and (self.adam.touches(dw) or (db is not None and self.adam.touches(db))):
                raise RuntimeError('fuse_adam: a weight gradient receives a contribution after a fused launch has '
                                   'already applied Adam to it (a layer used twice per step): disable MVAE_FUSE_ADAM')

        def issue():
            fused = (adam is not None and items
                     and all((it[0] is None) or (not it[4] and key_of.get(it[2].data_ptr()) in fusable) for it in items))
            for it in items:
                if it[0] is not None:
                    guard(it[2], it[3])
            if fused:
                for it in items:
                    adam.cover(it[2])
                    if it[3] is not None:
                        adam.cover(it[3])
                K.linear_wgrad_batched(items, adam=adam.struct)
            else:
                real = [it for it in items if it[0] is not None]      # finished gradients wait for rest()
                if len(real) == 1:
                    K.linear_wgrad(*real[0][:4], accumulate=real[0][4])
                elif real:
                    K.linear_wgrad_batched(real)
            del items[:]
            seen.clear()

        for e, key in zip(self, keys):
            if not isinstance(e, tuple):
                e()
                continue
            _, op, g, x = e
            dw, db, acc = _lin_wgrad_targets(op)
            key_of[dw.data_ptr()] = key
            if dw.data_ptr() in seen:
                issue()                     # a second contribution to the same gradient: keep the order (grad_target made it accumulate)
            if K.wgrad_batchable(g, x):
                seen.add(dw.data_ptr())
                items.append((g, x, dw, db, acc))
            else:
                guard(dw, db)
                K.linear_wgrad(g, x, dw, db, accumulate=acc)
        if adam is not None and self._final:
            if any(it[4] or key_of.get(it[2].data_ptr()) not in fusable for it in items):
                issue()
            for grad in self._final:
                items.append((None, None, grad.reshape(-1), None, False))
            if not any(it[0] is not None for it in items):
                # nothing but finished gradients: they are the arena-wide launch's business (rest())
                del items[:]
        issue()
        del self[:]
        self._final = []


# ----------------------------------------------------------------------------- grouped executor
class GroupedPlans(object):
    """G stacks of IDENTICAL structure -- Embedding / Linear / paired heads / Swish only -- executed
    one grouped launch per layer (celeba19's 18 attribute encoders and decoders,
    celeba19/model.py:29-30).  Works on the experts' slices of the parameter arena: layer j of
    expert g must sit at a uniform stride from layer j of expert 0, which the arena's module
    order guarantees for identical sub-modules laid out back to back."""

    def __init__(self, plans):
        self.plans = [list(pl) for pl in plans]
        self.G = len(self.plans)
        p0 = self.plans[0]
        for pl in self.plans:
            if len(pl) != len(p0):
                raise RuntimeError('grouped stacks differ in depth')
            for a, b in zip(pl, p0):
                if a.kind != b.kind or a.act != b.act or a.drop != 0 or a.kind not in ('emb', 'lin', 'lin2'):
                    raise RuntimeError('grouped stacks must be identical Embedding/Linear/Swish chains')
        if p0[-1].act:
            raise RuntimeError('grouped stacks must end in a plain Linear')

    def _params(self, j):
        """per expert: (weight, bias or None) tensors of layer j (joined views for paired heads)"""
        out = []
        for pl in self.plans:
            op = pl[j]
            out.append((op.mod.weight, None) if op.kind == 'emb' else _lin_weights(op))
        return out

    @staticmethod
    def _stride(ts):
        if ts[0] is None:
            return 0
        if len(ts) == 1:
            return 0
        d = (ts[1].data_ptr() - ts[0].data_ptr()) // 4
        for g, t in enumerate(ts):
            if t.shape != ts[0].shape or t.data_ptr() - ts[0].data_ptr() != 4 * d * g:
                raise RuntimeError('grouped stacks need uniformly strided parameters (one arena, identical '
                                   'experts laid out back to back)')
        if d <= 0:
            raise RuntimeError('grouped parameter stride must be positive')
        return d

    def layer(self, j):
        """(w0, w_stride, b0, b_stride) of layer j -- data tensors"""
        ps = self._params(j)
        ws, bs = [w for w, _ in ps], [b for _, b in ps]
        return ws[0].detach(), self._stride(ws), (None if bs[0] is None else bs[0].detach()), self._stride(bs)

    def grads(self, j):
        """Gradient destinations of layer j for all experts: (dw0, db0 or None, accumulate)."""
        accs = set()
        for pl in self.plans:
            op = pl[j]
            ps = list(op.mod.parameters()) if op.kind != 'lin2' else [q for h in op.mod.heads for q in h.parameters()]
            for q in ps:
                accs.add(grad_target(q)[1])
        if len(accs) != 1:
            raise RuntimeError('grouped gradients out of sync')
        op = self.plans[0][j]
        if op.kind == 'lin2':
            a, b = op.mod.heads
            arena = a.weight._arena
            return arena.joined(a.weight, b.weight)[1], arena.joined(a.bias, b.bias)[1], accs.pop()
        return op.mod.weight.grad, (op.mod.bias.grad if getattr(op.mod, 'bias', None) is not None else None), accs.pop()


def forward_tape_grouped(gp, x, final_out=None):
    """x: float [rows, G] index columns when the stacks start with an Embedding, else
    [G, rows, width].  Returns (output [G, rows, N], tape)."""
    tape, h = [], x
    G = gp.G
    last = len(gp.plans[0]) - 1
    for j, op in enumerate(gp.plans[0]):
        w0, w_gs, b0, b_gs = gp.layer(j)
        if op.kind == 'emb':
            act = torch.empty(G, h.shape[0], w0.shape[1], dtype=torch.float32, device=h.device)
            K.embedding_swish_fwd_grouped(h, w0, w_gs, act)
            tape.append((h, None))
            h = act
            continue
        M, N = h.shape[1], w0.shape[0]
        if j == last and final_out is not None:
            pre = final_out.reshape(G, M, N)
        else:
            pre = torch.empty(G, M, N, dtype=torch.float32, device=h.device)
        act = torch.empty(G, M, N, dtype=torch.float32, device=h.device) if op.act else None
        K.linear_fwd_grouped(h, w0, w_gs, b0, b_gs, pre, act)
        tape.append((h, pre if op.act else None))
        h = act if op.act else pre
    return h, tape


def backward_tape_grouped(gp, tape, g, need_input_grad=False, input_grad_out=None):
    """g: [G, rows, N] gradient of the stacks' outputs.  Parameter gradients go to the arena."""
    plan = gp.plans[0]
    for j in range(len(plan) - 1, -1, -1):
        op = plan[j]
        x, _ = tape[j]
        w0, w_gs, _, b_gs = gp.layer(j)
        dw0, db0, acc = gp.grads(j)
        if op.kind == 'emb':
            K.embedding_swish_bwd_grouped(x, w0, w_gs, g.contiguous(), dw0, accumulate=acc)
            return None
        K.linear_wgrad_grouped(g, x, dw0, w_gs, db0, b_gs, accumulate=acc)
        if j > 0 or need_input_grad:
            pre_in = tape[j - 1][1] if (j > 0 and plan[j - 1].kind != 'emb') else None
            if j == 0 and input_grad_out is not None:
                dx = input_grad_out
            else:
                dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            K.linear_dgrad_grouped(g, w0, w_gs, dx, pre_in)
            g = dx
    return g if need_input_grad else None


# ----------------------------------------------------------------------------- autograd bridge
class StackFn(torch.autograd.Function):
    """autograd wrapper used by the module surface.  Parameters are passed as inputs only so
    that autograd schedules the backward; their gradients are written to the arena by the
    kernels and ``None`` is returned for them."""

    @staticmethod
    def forward(ctx, x, holder, *params):
        plan, groups, masks, bn_updates = holder
        out, tape = forward_tape(plan, x, groups=groups, masks=masks, bn_updates=bn_updates, training=True)
        ctx.plan, ctx.tape, ctx.groups = plan, tape, groups
        ctx.n_params = len(params)
        ctx.x_needs_grad = bool(ctx.needs_input_grad[0])
        return out

    @staticmethod
    def backward(ctx, g):
        gx = backward_tape(ctx.plan, ctx.tape, g, need_input_grad=ctx.x_needs_grad, groups=ctx.groups)
        ctx.tape = None
        return (gx, None) + (None,) * ctx.n_params


def run_plan(plan, x, groups=1, masks=None, bn_updates=1, training=True):
    """Module-surface entry: autograd-tracked in training mode, plain launches otherwise."""
    if not x.is_cuda:
        raise RuntimeError('multimodal-vae-public_amd runs on the GPU only (input on %s): move the model '
                           'and the batch to cuda; there is no CPU fallback' % x.device)
    if training and torch.is_grad_enabled():
        return StackFn.apply(x, (plan, groups, masks, bn_updates), *plan_params(plan))
    out, _ = forward_tape(plan, x, groups=groups, masks=masks, bn_updates=bn_updates, training=training)
    return out


def run_stack(mods, x):
    mods = list(mods)
    training = any(m.training for m in mods)
    return run_plan(compile_plan(mods), x, training=training)

"""FusedAdam: ``torch.optim.Adam`` (default hyper-parameters, as the reference uses it --
mnist/train.py:168,219) as ONE HIP launch over the flat parameter arena.

The reference's optimizer touches 25-260 parameter tensors with ~10 ATen kernels each; with
every parameter and gradient contiguous in the arena the update is a single streaming pass
(28 bytes / parameter, HBM-bound).  The step counter lives on the device and is advanced by the
kernel, so the optimizer step can sit inside a captured hipGraph.

Drop-in surface: ``FusedAdam(model.parameters(), lr=...)``, ``zero_grad()``, ``step()``,
``state_dict()`` / ``load_state_dict()`` with torch.optim.Adam's layout
(``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}``), so checkpoints interchange
(mnist/train.py:263-268 stores ``optimizer.state_dict()``).
"""
import torch

from . import _lib
from . import kernels as K


class AdamFusion(object):
    """One step's record of the parameters that weight-gradient launches have already updated themselves
    (``K.linear_wgrad_batched(items, adam=fusion.struct)``; include/mvae_hip.h: mvae_linear_wgrad_batched_adam).
    ``cover(grad)`` is called for every gradient tensor handed to such a launch; ``rest()`` is what
    ``FusedAdam.step_counted(fusion)`` still has to update -- for an all-Linear model (mnist/model.py) nothing:
    optimizer.step() has then no launch of its own at the end of the backward chain."""

    def __init__(self, opt):
        arena, g = opt._arena, opt.param_groups[0]
        self.arena = arena
        self.coef = torch.zeros(2, dtype=torch.float32, device=arena.flat.device)
        self.struct = _lib.AdamFuse(arena.grad.data_ptr(), arena.flat.data_ptr(), opt._m.data_ptr(), opt._v.data_ptr(),
                                    self.coef.data_ptr(), g['betas'][0], g['betas'][1], g['eps'], opt.grad_scale)
        self._keep = (opt._m, opt._v)        # the struct holds raw addresses
        self.covered = []

    def begin(self):
        self.covered = []

    def cover(self, grad):
        base = self.arena.grad.data_ptr()
        lo = (grad.data_ptr() - base) // 4
        hi = lo + grad.numel()
        if (grad.data_ptr() - base) % 4 or lo < 0 or hi > self.arena.numel or not grad.is_contiguous():
            raise RuntimeError('a fused update needs a contiguous slice of the gradient arena')
        for a, b in self.covered:
            if lo < b and a < hi:
                raise RuntimeError('arena elements [%d, %d) would be updated twice in one step' % (max(lo, a), min(hi, b)))
        self.covered.append((lo, hi))

    def touches(self, grad):
        """True when ``grad`` overlaps a range a fused launch has ALREADY updated this step: a further contribution
        to it would arrive after its Adam update (and ``rest()`` would skip the parameter)."""
        lo = (grad.data_ptr() - self.arena.grad.data_ptr()) // 4
        hi = lo + grad.numel()
        return any(lo < b and a < hi for a, b in self.covered)

    def rest(self):
        """Arena ranges (whole parameters, merged) no fused launch has updated this step."""
        out, cur = [], None
        for p, o in zip(self.arena.params, self.arena.offsets):
            n = p.numel()
            inside = [(a, b) for a, b in self.covered if a < o + n and o < b]
            if inside:
                # a parameter is covered by one launch item, or by the two halves of nothing: never partially
                if not any(a <= o and o + n <= b for a, b in inside):
                    raise RuntimeError('a parameter was only partially updated by a fused launch')
                if cur is not None:
                    out.append(tuple(cur))
                    cur = None
            elif cur is None:
                cur = [o, o + n]
            else:
                cur[1] = o + n
        if cur is not None:
            out.append(tuple(cur))
        return out


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self.grad_scale = float(grad_scale)   # 1/world_size when gradients were all-reduced with SUM
        self._arena = None
        self._m = self._v = self._step_dev = None

    def _bind(self):
        ps = [p for g in self.param_groups for p in g['params']]
        arena = getattr(ps[0], '_arena', None)
        if arena is None:
            raise RuntimeError('FusedAdam needs arena-backed parameters: call model.cuda() and '
                               'model.finalize() (or run one forward) before the first step')
        for p in ps:
            if getattr(p, '_arena', None) is not arena:
                raise RuntimeError('all parameters must live in one arena')
        if len(ps) != len(arena.params):
            raise RuntimeError('FusedAdam updates the whole arena; pass model.parameters()')
        self._arena = arena
        self._m = torch.zeros_like(arena.flat)
        self._v = torch.zeros_like(arena.flat)
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=arena.flat.device)
        self._host_step = 0
        self._fusion = None
        self._coef = torch.zeros(2, dtype=torch.float32, device=arena.flat.device)
        self._coef_ready = False

    def zero_grad(self, set_to_none=True):
        # None marks "first write of the step overwrites": no memset over the gradient arena
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError('closures are not supported')
        arena = self._arena
        if arena is None or arena is not getattr(self.param_groups[0]['params'][0], '_arena', None):
            self._bind()
            arena = self._arena
        for p in arena.params:
            if p.grad is None:
                raise RuntimeError('a parameter received no gradient this step')
        g = self.param_groups[0]
        K.adam_step(arena.flat, arena.grad, self._m, self._v, self._step_dev, g['lr'], g['betas'][0],
                    g['betas'][1], g['eps'], self.grad_scale)
        self._host_step += 1

    def step_counter(self):
        """The device step counter (binding the arena if needed): a fused step engine advances it itself early
        in the step, off the critical chain, and then calls ``step_counted()`` instead of ``step()``."""
        arena = self._arena
        if arena is None or arena is not getattr(self.param_groups[0]['params'][0], '_arena', None):
            self._bind()
        return self._step_dev

    def fusion(self):
        """The ``AdamFusion`` of this optimizer (one per arena binding): a fused step engine hands it to its
        weight-gradient launches, advances the counter with ``prepare_counted`` and ends the step with
        ``step_counted(fusion)``."""
        arena = self._arena
        if arena is None or arena is not getattr(self.param_groups[0]['params'][0], '_arena', None):
            self._bind()
        if self._fusion is None:
            self._fusion = AdamFusion(self)
        return self._fusion

    def prepare_counted(self, fusion):
        """Advance the step counter by one AND leave the step's two bias-correction factors where the fused
        weight-gradient launches read them (instead of ``K.counter_add(step_counter(), 1)``)."""
        g = self.param_groups[0]
        K.adam_prepare(self._step_dev, 1, g['lr'], g['betas'][0], g['betas'][1], fusion.coef)

    def prepare_plain(self):
        """The counter launch of a fused step WITHOUT fused weight-gradient updates: advance the counter by one and leave
        the step's two bias-correction factors for ``step_counted()``, whose launch then starts streaming at once
        (mvae_adam_apply_coef) instead of computing two double-precision powers per wavefront first."""
        g = self.param_groups[0]
        K.adam_prepare(self.step_counter(), 1, g['lr'], g['betas'][0], g['betas'][1], self._coef)
        self._coef_ready = True

    def _apply_counted(self, lo, hi):
        arena, g = self._arena, self.param_groups[0]
        if self._coef_ready:
            K.adam_apply_coef(arena.flat[lo:hi], arena.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], self._coef,
                              g['betas'][0], g['betas'][1], g['eps'], self.grad_scale)
        else:
            K.adam_apply_at(arena.flat[lo:hi], arena.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], self._step_dev, 0,
                            g['lr'], g['betas'][0], g['betas'][1], g['eps'], self.grad_scale)

    @torch.no_grad()
    def step_counted(self, fusion=None):
        """``step()`` for a caller that already advanced ``step_counter()`` by one this step: one launch -- or, with
        the step's ``AdamFusion``, one launch per arena range its weight-gradient launches have not updated."""
        arena = self._arena
        for p in arena.params:
            if p.grad is None:
                raise RuntimeError('a parameter received no gradient this step')
        for lo, hi in ([(0, arena.numel)] if fusion is None else fusion.rest()):
            self._apply_counted(lo, hi)
        self._coef_ready = False
        self._host_step += 1

    @torch.no_grad()
    def step_counted_range(self, lo, hi, last=False):
        """``step_counted()`` in pieces: Adam on arena elements [lo, hi) at t = the (already advanced) counter.  A fused
        step updates its decoders' range as soon as their weight gradients are final -- beside the encoders' backward --
        and only the encoders' range at the end of the chain; ``last`` closes the step on the host side."""
        arena = self._arena
        self._apply_counted(lo, hi)
        if last:
            for p in arena.params:
                if p.grad is None:
                    raise RuntimeError('a parameter received no gradient this step')
            self._coef_ready = False
            self._host_step += 1

    @torch.no_grad()
    def step_range(self, lo, hi):
        """Adam on arena elements [lo, hi) at the CURRENT step (counter not advanced): data-parallel
        replicas call this per gradient bucket as its all-reduce lands, then ``advance()`` once."""
        arena = self._arena
        if arena is None or arena is not getattr(self.param_groups[0]['params'][0], '_arena', None):
            self._bind()
            arena = self._arena
        g = self.param_groups[0]
        K.adam_apply(arena.flat[lo:hi], arena.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], self._step_dev, g['lr'],
                     g['betas'][0], g['betas'][1], g['eps'], self.grad_scale)

    def advance(self):
        K.counter_add(self._step_dev, 1)
        self._host_step += 1

    # torch.optim.Adam-compatible checkpoint layout
    def state_dict(self):
        if self._arena is not None:
            self._host_step = int(self._step_dev.item())   # graph replays advance only the device counter
            step = torch.tensor(float(self._host_step))
            for p in self._arena.params:
                o = p._arena_off
                self.state[p] = {'step': step.clone(),
                                 'exp_avg': self._m[o:o + p.numel()].view(p.shape),
                                 'exp_avg_sq': self._v[o:o + p.numel()].view(p.shape)}
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self._bind()
        step = 0
        for p in self._arena.params:
            st = self.state.get(p)
            if st:
                o = p._arena_off
                self._m[o:o + p.numel()].copy_(st['exp_avg'].reshape(-1))
                self._v[o:o + p.numel()].copy_(st['exp_avg_sq'].reshape(-1))
                step = int(st['step'])
        self._host_step = step
        self._step_dev.fill_(step)

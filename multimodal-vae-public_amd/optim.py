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

from . import kernels as K


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

    @torch.no_grad()
    def step_counted(self):
        """``step()`` for a caller that already advanced ``step_counter()`` by one this step: one launch."""
        arena = self._arena
        for p in arena.params:
            if p.grad is None:
                raise RuntimeError('a parameter received no gradient this step')
        g = self.param_groups[0]
        K.adam_apply_at(arena.flat, arena.grad, self._m, self._v, self._step_dev, 0, g['lr'], g['betas'][0],
                        g['betas'][1], g['eps'], self.grad_scale)
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

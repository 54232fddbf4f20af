"""``capture_step``: the reference's train-loop body, UNCHANGED, as one hipGraph.

The fused engines (``engine.BimodalStep`` / ``Celeba19Step``) restructure the step (each encoder once, loss folded into
the decoders, two streams).  A user who keeps the reference's loop --

    optimizer.zero_grad()
    recon_image_1, recon_text_1, mu_1, logvar_1 = model(image, text)          # mnist/train.py:197-219
    recon_image_2, recon_text_2, mu_2, logvar_2 = model(image)
    recon_image_3, recon_text_3, mu_3, logvar_3 = model(text=text)
    joint_loss = elbo_loss(recon_image_1, image, recon_text_1, text, mu_1, logvar_1, ..., annealing_factor=annealing_factor)
    ...
    train_loss = joint_loss + image_loss + text_loss
    train_loss.backward()
    optimizer.step()

-- on the drop-in modules gets correct numbers but pays ~150 ctypes launches, autograd shells and allocator calls per step
on the host (MNIST B = 512: 3.5 ms against 0.29 ms for the fused engine, BENCH_r04 ``module_surface``).  Wrapping that body
in a closure and handing it to ``capture_step`` removes the host from the step: one eager pass to warm the allocator, one
capture (forward, autograd backward, optimizer), then every call is input copies + ONE graph launch.

    def body(image, text, annealing_factor):           # the lines above, verbatim
        ...
        return train_loss
    step = mvae_amd.capture_step(body, (image, text, 1.0), model=model, optimizer=optimizer)
    for image, text in loader:
        loss = step(image, text, annealing_factor)      # same tensor object every call: read it (.item()) before the next

What makes the unmodified body capturable: every kernel behind ``model()`` / ``elbo_loss`` is an enqueue on the current
stream; noise comes from the device-side Philox counter (``base.MVAE.device_randn``); python numbers among the arguments
(the annealing factor) become 0-d device tensors that the graph re-reads at each replay (``functional._w``); BatchNorm's
``num_batches_tracked`` is advanced on the host per replay by what one step adds.  The optimizer must be capturable:
``optim.FusedAdam`` (device step counter) or ``torch.optim.Adam(..., capturable=True)``.
Shapes are frozen at capture: a last, shorter batch of an epoch runs the body eagerly (``step.eager(...)``).

What else is frozen: the graph holds ADDRESSES and by-value kernel arguments.  The optimizer's hyper-parameters
(lr, betas, eps, ...) went into the captured launches as numbers -- a scheduler or a hand edit of ``param_groups`` afterwards
would be silently ignored by replays -- and the model's device noise counter (``_rng``) is the tensor object that existed at
capture: ``model.seed_noise()`` replaces it, and replays would keep drawing from the old one while ``step.eager()`` used the
new.  Both are checked per call: a changed hyper-parameter or a replaced counter raises (capture again, or reseed in place
with ``step.reseed(seed)``).
"""
import torch

from . import layers as L

__all__ = ['capture_step', 'CapturedStep']


def _hyper_snapshot(optimizer):
    """The by-value hyper-parameters of every param group (everything that is not the parameter list or a tensor)."""
    snap = []
    for g in optimizer.param_groups:
        snap.append(tuple(sorted((k, repr(v)) for k, v in g.items() if k != 'params' and not torch.is_tensor(v))))
    return tuple(snap)


def _optimizer_tensors(optimizer):
    out = []
    for k in ('_m', '_v', '_step_dev'):                    # FusedAdam's flat state
        t = getattr(optimizer, k, None)
        if torch.is_tensor(t):
            out.append(t)
    for st in optimizer.state.values():                     # stock optimizers: per-parameter state
        for v in st.values():
            if torch.is_tensor(v):
                out.append(v)
    return out


class CapturedStep(object):
    def __init__(self, fn, example_args, model, optimizer, warmup=2):
        if not torch.cuda.is_available():
            raise RuntimeError('capture_step needs the GPU (there is no CPU path)')
        self.fn, self.model, self.optimizer = fn, model, optimizer
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('move the model to the GPU first')
        if isinstance(optimizer, torch.optim.Adam) and not optimizer.defaults.get('capturable', False):
            raise RuntimeError('torch.optim.Adam keeps its step count on the host: pass capturable=True, or use mvae_amd.optim.FusedAdam')
        self.static = []
        for a in example_args:
            if torch.is_tensor(a):
                self.static.append(a.detach().to(dev).clone())
            elif isinstance(a, (int, float)) and not isinstance(a, bool):
                self.static.append(torch.full((), float(a), dtype=torch.float32, device=dev))
            elif a is None:
                self.static.append(None)
            else:
                raise TypeError('capture_step arguments are tensors, python numbers or None (got %s)' % type(a).__name__)
        if hasattr(model, 'finalize'):
            model.finalize()
        bns = [m for m in model.modules() if isinstance(m, L._BatchNormMixin)]
        # ---- everything the warm-up passes mutate, to be put back: the capture must leave no trace
        params = [p.detach().clone() for p in model.parameters()]
        bufs = [b.detach().clone() for b in model.buffers()]
        pend = [m._nbt_pending for m in bns]
        rng = model.__dict__.get('_rng')
        rng_ctr = None if rng is None else rng[1].clone()
        # optimizer state is updated in place (FusedAdam, capturable Adam): a tensor that exists now is put back, one the
        # warm-up creates is what a fresh optimizer would hold -- zeros
        opt_state = {id(t): (t, t.detach().clone()) for t in _optimizer_tensors(optimizer)}
        host_step = getattr(optimizer, '_host_step', None)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                  # allocator warm-up + lazily created state (optimizer moments, arena views)
                before = [m._nbt_pending for m in bns]
                fn(*self.static)
                self._bn_inc = [(m, m._nbt_pending - b) for m, b in zip(bns, before)]
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn(*self.static)
        self._hyper = _hyper_snapshot(optimizer)
        self._rng_obj = model.__dict__.get('_rng')
        # ---- back to the state before the first warm-up pass (in place: the graph holds these addresses)
        with torch.no_grad():
            for p, s in zip(model.parameters(), params):
                p.copy_(s)
            for b, s in zip(model.buffers(), bufs):
                b.copy_(s)
            for m, n in zip(bns, pend):
                m._nbt_pending = n
            if rng_ctr is not None:
                rng[1].copy_(rng_ctr)
            elif model.__dict__.get('_rng') is not None:
                model.__dict__['_rng'][1].zero_()
            for t in _optimizer_tensors(optimizer):
                saved = opt_state.get(id(t))
                if saved is not None and saved[0] is t:
                    t.copy_(saved[1])
                else:
                    t.zero_()
            if host_step is not None:
                optimizer._host_step = host_step
        torch.cuda.synchronize(dev)

    def __call__(self, *args):
        if len(args) != len(self.static):
            raise TypeError('captured with %d arguments, called with %d' % (len(self.static), len(args)))
        for s, a in zip(self.static, args):
            if s is None:
                if a is not None:
                    raise TypeError('an argument captured as None must stay None')
            elif torch.is_tensor(a):
                if a.shape != s.shape:
                    raise ValueError('captured for shape %s, called with %s: run step.eager(...) for a ragged last batch'
                                     % (tuple(s.shape), tuple(a.shape)))
                s.copy_(a, non_blocking=True)
            else:
                s.fill_(float(a))
        if _hyper_snapshot(self.optimizer) != self._hyper:
            raise RuntimeError('the optimizer\'s hyper-parameters changed after capture_step: the graph holds the old values '
                               '(capture the step again)')
        if self.model.__dict__.get('_rng') is not self._rng_obj:
            raise RuntimeError('the model\'s noise counter was replaced after capture_step (seed_noise): the graph draws from '
                               'the old one -- capture again, or reseed in place with step.reseed(seed)')
        for m, inc in self._bn_inc:
            m._nbt_pending += inc
        self.graph.replay()
        return self.out

    def reseed(self, seed):
        """Restart the captured step's noise stream IN PLACE (same counter tensor, so the graph sees it): the Philox key of
        the captured launches is fixed at capture; the counter is set from ``seed`` so that different seeds draw disjoint
        streams."""
        rng = self._rng_obj
        if rng is None:
            raise RuntimeError('this model draws no device noise')
        rng[1].fill_((int(seed) & 0x7fffffff) << 32)

    def eager(self, *args):
        """The same body without the graph (any batch size)."""
        return self.fn(*args)


def capture_step(fn, example_args, model, optimizer, warmup=2):
    """Capture ``fn(*example_args)`` -- a whole train step: zero_grad, forward, loss, backward, optimizer.step -- into one
    hipGraph and return the callable that replays it (see the module docstring)."""
    return CapturedStep(fn, tuple(example_args), model, optimizer, warmup=warmup)

"""Flat parameter / gradient arena.

All parameters of an MVAE live in ONE contiguous fp32 buffer in HBM and all gradients in a
second one of the same layout.  ``p.data`` / ``p.grad`` of every ``nn.Parameter`` are views, so
the reference's surface (``model.parameters()``, ``state_dict()``, ``torch.optim.Adam``) keeps
working, while
  * the weight-gradient kernels write straight into the gradient arena (no per-parameter
    AccumulateGrad adds),
  * Adam is one launch over the whole arena (``optim.FusedAdam``),
  * data-parallel replicas all-reduce a few large contiguous buckets (``parallel.py``) instead
    of one collective per tensor -- sized for xGMI, not for a per-parameter hook storm.
The layout follows backward-completion order (decoders first, then encoders) so bucket k is
final while the backward of bucket k+1 is still running.
"""
import torch

ALIGN = 4  # floats: every parameter starts 16-byte aligned for float4 loads


class ParamArena(object):
    def __init__(self, module, order=None, adjacent=(), tail=()):
        """``order``: sub-modules in the order their gradients complete during backward
        (default: registration order).  ``adjacent``: tuples of parameters that must be laid
        out back to back (e.g. the mu / logvar heads of the MNIST encoders, so that both heads
        are one GEMM).  ``tail``: modules whose parameters go to the very END of the arena -- the layers
        whose gradients complete last (the image encoder's first layers): data-parallel replicas all-reduce
        them as a small last bucket, ``tail_range``."""
        params = []
        seen = set()
        mods = list(order) if order is not None else [module]
        for m in mods + [module]:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        tail_ids = set(id(p) for m in tail for p in m.parameters())
        params = [p for p in params if id(p) not in tail_ids] + [p for p in params if id(p) in tail_ids]
        follow = {}
        for tup in adjacent:
            for a, b in zip(tup[:-1], tup[1:]):
                follow[id(a)] = b
        placed, ordered = set(), []
        tails = set(id(b) for b in follow.values())
        for p in params:
            if id(p) in placed or id(p) in tails:
                continue
            q = p
            while q is not None and id(q) not in placed:
                ordered.append(q)
                placed.add(id(q))
                q = follow.get(id(q))
        for p in params:           # tails whose head never showed up (defensive)
            if id(p) not in placed:
                ordered.append(p)
                placed.add(id(p))
        device = ordered[0].device
        offsets, off = [], 0
        for p in ordered:
            if p.dtype != torch.float32 or p.device != device:
                raise RuntimeError('arena needs fp32 parameters on one device')
            offsets.append(off)
            n = p.numel()
            nxt = follow.get(id(p))
            off += n if nxt is not None else (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.params = ordered
        self.offsets = offsets
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.module_ranges = {}
        with torch.no_grad():
            for p, o in zip(ordered, offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._arena = self
                p._arena_off = o
                p.grad = None
        if order is not None:
            for m in order:
                offs = [p._arena_off for p in m.parameters() if id(p) not in tail_ids]
                ends = [p._arena_off + p.numel() for p in m.parameters() if id(p) not in tail_ids]
                if offs:
                    self.module_ranges[m] = (min(offs), max(ends))     # without the module's tail parameters
        self.tail_range = None
        if tail_ids:
            offs = [p._arena_off for p in ordered if id(p) in tail_ids]
            self.tail_range = (min(offs), self.numel)
            if any(id(p) not in tail_ids for p in ordered if p._arena_off >= self.tail_range[0]):
                raise RuntimeError('tail parameters are not contiguous at the end of the arena')

    def grad_view(self, p):
        o = p._arena_off
        return self.grad[o:o + p.numel()].view(p.shape)

    def attach_grads(self):
        """Point every ``p.grad`` at its slice of the gradient arena."""
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * p._arena_off:
                p.grad = self.grad_view(p)

    def joined(self, first, second):
        """View of two adjacent parameters as one tensor stacked along dim 0."""
        if first._arena_off + first.numel() != second._arena_off:
            raise RuntimeError('parameters are not adjacent in the arena')
        n = first.numel() + second.numel()
        shape = (first.shape[0] + second.shape[0],) + tuple(first.shape[1:])
        o = first._arena_off
        return self.flat[o:o + n].view(shape), self.grad[o:o + n].view(shape)


def grad_target(p):
    """Where a weight-gradient kernel should write for parameter ``p`` and whether it must
    accumulate.  ``p.grad is None`` (after ``zero_grad()``) means the first write of this step
    overwrites -- no memset pass over the gradients is ever needed."""
    if p.grad is None:
        arena = getattr(p, '_arena', None)
        p.grad = arena.grad_view(p) if arena is not None else torch.empty_like(p.data)
        return p.grad, False
    return p.grad, True

"""Data-parallel replicas: one process per GPU, gradients averaged with RCCL over xGMI.

The reference is single-device; this is the new component north_star asks
for.  Design for MI355X rather than a per-parameter DDP hook storm:

  * gradients already live in ONE flat arena laid out in backward-completion order, so the
    exchange is a few large contiguous all-reduces (bucket = the decoders' range, then the
    encoders' range; 10-90 MB total) -- large messages are what the 7 x 153 GB/s point-to-point
    xGMI links want, and RCCL picks the direct all-to-all reduce-scatter/all-gather schedule on
    the fully connected 8-GPU mesh for them;
  * bucket k is launched (``async_op``) from the fused step the moment the last weight-gradient
    kernel of its range has been enqueued; RCCL runs it on its own stream behind an event, so it
    overlaps with the remaining backward; ``wait()`` fences before the optimizer;
  * the sum is scaled by 1/N inside FusedAdam's read of the gradient (no extra pass);
  * BatchNorm statistics stay per replica, exactly as N independent reference processes would
    (no sync-BN in the reference); parameters and buffers are broadcast from rank 0 once.

``GradBuckets`` is backend-agnostic (it only needs a flat tensor), which is how the gloo CPU
tests exercise the N > 1 logic without GPUs.
"""
import torch.distributed as dist


class GradBuckets(object):
    """Split ``flat_grad`` into contiguous [lo, hi) buckets and all-reduce them asynchronously."""

    def __init__(self, flat_grad, ranges, group=None):
        self.flat = flat_grad
        self.ranges = [(int(lo), int(hi)) for lo, hi in ranges]
        lo_all = min(r[0] for r in self.ranges)
        hi_all = max(r[1] for r in self.ranges)
        covered = sorted(self.ranges)
        pos = lo_all
        for lo, hi in covered:
            if lo != pos:
                raise ValueError('bucket ranges must tile the gradient arena without gaps')
            pos = hi
        if lo_all != 0 or hi_all != flat_grad.numel():
            raise ValueError('bucket ranges must cover the whole arena')
        self.group = group
        self.pending = []

    def launch(self, k):
        lo, hi = self.ranges[k]
        work = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append(work)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


class DataParallel(object):
    """Wrap a fused step engine: ``dp = DataParallel(model, engine)``; then every
    ``engine.forward_backward`` launches the bucket all-reduces itself and ``dp.wait()`` must be
    called before ``optimizer.step()`` (``FusedAdam(..., grad_scale=dp.grad_scale)``)."""

    def __init__(self, model, engine, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.world = dist.get_world_size(group)
        self.grad_scale = 1.0 / self.world
        arena = model.finalize()
        dist.broadcast(arena.flat, src=0, group=group)
        for b in model.buffers():
            dist.broadcast(b, src=0, group=group)
        self.buckets = GradBuckets(arena.grad, bucket_ranges(model, arena), group=group)
        engine.on_bucket_ready = self.buckets.launch
        self.engine = engine

    def launch(self, k):
        """Start the all-reduce of bucket k (called by the engine between captured graphs)."""
        self.buckets.launch(k)

    def wait(self):
        self.buckets.wait()


def bucket_ranges(model, arena):
    """Two buckets in backward-completion order: all decoders, then all encoders."""
    order = model.arena_order()
    n_dec = sum(1 for m in order if 'Decoder' in type(m).__name__)
    dec = [arena.module_ranges[m] for m in order[:n_dec]]
    enc = [arena.module_ranges[m] for m in order[n_dec:]]
    if not enc:
        return [(0, arena.numel)]
    split = min(lo for lo, _ in enc)       # first encoder parameter (16-byte aligned start)
    if max(hi for _, hi in dec) > split:
        raise RuntimeError('arena layout is not decoders-then-encoders')
    return [(0, split), (split, arena.numel)]

"""Data-parallel replicas: one process per GPU, gradients averaged with RCCL over xGMI.

The reference is single-device; this is the new component north_star asks
for.  Design for MI355X rather than a per-parameter DDP hook storm:

  * gradients already live in ONE flat arena laid out in backward-completion order, so the
    exchange is two or three large contiguous all-reduces (the decoders' range, the
    encoders' range, and -- for the conv models -- the image encoder's first layers as a small last
    bucket; 10-90 MB total) -- large messages are what the 7 x 153 GB/s point-to-point
    xGMI links want, and RCCL picks the direct all-to-all reduce-scatter/all-gather schedule on
    the fully connected 8-GPU mesh for them;
  * bucket k is launched (``async_op``) from the fused step the moment the last weight-gradient
    kernel of its range has been enqueued; RCCL runs it on its own stream behind an event, so it
    overlaps with the remaining backward;
  * Adam runs per bucket as each reduction lands (``finish``): the last collective -- the image
    encoder's first layers, a few MB at most -- is the only one not hidden behind backward work, and
    the update of the earlier buckets hides most of it;
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
        self.pending = {}

    def launch(self, k):
        lo, hi = self.ranges[k]
        if k in self.pending:
            raise RuntimeError('bucket %d launched twice in one step' % k)
        self.pending[k] = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, k=None):
        """Fence bucket ``k`` (all pending ones when None).  With the nccl (= RCCL) backend this makes the
        CURRENT STREAM wait for the collective -- the host does not block; gloo blocks the host."""
        keys = sorted(self.pending) if k is None else [k]
        for key in keys:
            work = self.pending.pop(key, None)      # already fenced: nothing to do
            if work is not None:
                work.wait()


class DataParallel(object):
    """Wrap a fused step engine: ``dp = DataParallel(model, engine)``.  Every
    ``engine.forward_backward`` then launches bucket k's all-reduce the moment the last weight gradient of
    its range has been enqueued, and ``dp.finish(optimizer)`` applies Adam bucket by bucket as the
    reductions land (``FusedAdam(..., grad_scale=dp.grad_scale)``): the update of bucket k overlaps the
    all-reduce of bucket k+1, and the last bucket -- the image encoder's first layers -- is small."""

    def __init__(self, model, engine, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.world = dist.get_world_size(group)
        self.grad_scale = 1.0 / self.world
        arena = model.finalize()
        dist.broadcast(arena.flat, src=0, group=group)
        for b in model.buffers():
            dist.broadcast(b, src=0, group=group)
        ranges = bucket_ranges(model, arena)
        self.buckets = GradBuckets(arena.grad, ranges, group=group)
        engine.configure_buckets(len(ranges))
        engine.on_bucket_ready = self.buckets.launch
        self.engine = engine

    @property
    def n_buckets(self):
        return len(self.buckets.ranges)

    def launch(self, k):
        """Start the all-reduce of bucket k (called by the engine between captured graphs)."""
        self.buckets.launch(k)

    def wait(self, k=None):
        self.buckets.wait(k)

    def finish(self, optimizer):
        """The optimizer step of a data-parallel replica: per bucket, fence its all-reduce and run Adam on
        its range (same step count for all ranges), then advance the step counter once."""
        for k, (lo, hi) in enumerate(self.buckets.ranges):
            self.buckets.wait(k)
            optimizer.step_range(lo, hi)
        optimizer.advance()


TAIL_SPLIT_MIN_BYTES = 8 << 20      # encoders smaller than this stay one bucket


def bucket_ranges(model, arena):
    """Buckets in backward-completion order: all decoders | the encoders without the image encoder's first
    layers | those first layers (``arena.tail_range``, laid out last).  Small models (MNIST: 4 MB of encoder
    parameters) keep the encoders in one bucket -- a third collective would cost more than it hides."""
    order = model.arena_order()
    n_dec = sum(1 for m in order if 'Decoder' in type(m).__name__)
    dec = [arena.module_ranges[m] for m in order[:n_dec]]
    enc = [arena.module_ranges[m] for m in order[n_dec:] if m in arena.module_ranges]
    if not enc:
        return [(0, arena.numel)]
    split = min(lo for lo, _ in enc)       # first encoder parameter (16-byte aligned start)
    if max(hi for _, hi in dec) > split:
        raise RuntimeError('arena layout is not decoders-then-encoders')
    tail = arena.tail_range
    if tail is not None and tail[0] > split and (arena.numel - split) * 4 >= TAIL_SPLIT_MIN_BYTES:
        return [(0, split), (split, tail[0]), (tail[0], arena.numel)]
    return [(0, split), (split, arena.numel)]

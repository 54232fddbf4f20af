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

Two transports for the bucket all-reduces:

  * ``RcclComm`` + ``RcclBuckets`` (default on GPUs, ``MVAE_COMM=rccl``): the library's own communicator behind
    the C ABI (``mvae_comm_*``, csrc/comm.hip -- what a host that is not PyTorch would call).  Enqueue-only stream /
    event operations + one RCCL enqueue, so the engine captures the WHOLE data-parallel step -- forward, backward,
    the bucket all-reduces on the communicator's stream, per-bucket Adam -- into ONE hipGraph: no host round-trip
    between buckets (the three-graph path cost MNIST +25 % at world size 1).  ``torch.distributed`` is then only
    the rendezvous that carries the 128-byte unique id.
  * ``GradBuckets`` over ``torch.distributed`` (``MVAE_COMM=torch``; always for gloo): ``async_op`` all-reduces
    issued between three captured graphs.
"""
import ctypes
import os
import sys

import torch
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

    def reset(self):
        """Forget launches a step that raised left behind (the next step would otherwise fail with 'launched twice'
        instead of the real error)."""
        self.pending.clear()

    def wait(self, k=None):
        """Fence bucket ``k`` (all pending ones when None).  With the nccl (= RCCL) backend this makes the
        CURRENT STREAM wait for the collective -- the host does not block; gloo blocks the host."""
        keys = sorted(self.pending) if k is None else [k]
        for key in keys:
            work = self.pending.pop(key, None)      # already fenced: nothing to do
            if work is not None:
                work.wait()


class RcclComm(object):
    """The library's RCCL communicator (``mvae_comm_*``): one per process, on the current device."""

    def __init__(self, rank, world, unique_id, device_index):
        from . import _lib
        self._lib = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(self._lib.mvae_comm_init(ctypes.byref(h), unique_id, len(unique_id), int(rank), int(world),
                                            int(device_index)), 'mvae_comm_init')
        self._h = h
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def bind_torch_rccl():
        """Point the C side at the librccl.so this process's torch already loaded: one RCCL copy per process."""
        from . import _lib
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        if os.path.exists(path) and not os.environ.get('MVAE_RCCL_LIB'):
            _lib.lib().mvae_comm_use_library(path.encode())      # refused (harmlessly) once a library is bound

    @classmethod
    def from_process_group(cls, device, group=None):
        """Collective over ``group``: rank 0 makes the unique id, torch.distributed carries it to the peers.

        Failure-safe rendezvous: every step that can fail on ONE rank is followed by a vote of ALL ranks
        before anyone enters the next collective, and every rank issues the same sequence of torch.distributed
        collectives whatever happened locally --

          1. each rank binds RCCL (dlopen + symbols + ncclGetVersion), rank 0 also creates the unique id; failures are
             caught, not raised;
          2. broadcast of [status byte | id] from rank 0 -- unconditional, also when rank 0 has no id to send;
          3. MIN all-reduce of "my binding worked and rank 0 had an id": if anyone failed, NOBODY calls
             ``ncclCommInitRank`` (a collective the failed rank would never join) and all raise the same error;
          4. ``mvae_comm_init`` on all ranks; then a second MIN vote, so a rank whose init failed does not leave its peers
             believing in a communicator it is not part of.
        """
        from . import _lib
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        why = None
        try:
            cls.bind_torch_rccl()
            if _lib.lib().mvae_comm_rccl_version() <= 0:
                why = 'RCCL could not be bound (dlopen / symbols / ncclGetVersion)'
        except Exception as e:
            why = '%s: %s' % (type(e).__name__, e)
        msg = torch.zeros(1 + _lib.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0 and why is None:
            buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
            rc = _lib.lib().mvae_comm_unique_id(buf, _lib.COMM_ID_BYTES)
            if rc == 0:
                msg[0] = 1
                msg[1:] = torch.tensor(list(buf), dtype=torch.uint8)
            else:
                why = 'mvae_comm_unique_id failed (%d)' % rc
        msg = msg.to(device)
        dist.broadcast(msg, src=0, group=group)
        host = msg.cpu()
        if why is None and int(host[0]) != 1:
            why = 'rank 0 could not create a unique id'
        cls._vote(why is None, device, group, why, 'binding RCCL / creating the unique id')
        raw = bytes(host[1:].tolist())
        if torch.device(device).type == 'cuda':
            torch.cuda.synchronize(device)
        comm, why = None, None
        try:
            comm = cls(rank, world, (ctypes.c_ubyte * len(raw)).from_buffer_copy(raw), torch.device(device).index or 0)
        except Exception as e:
            why = '%s: %s' % (type(e).__name__, e)
        try:
            cls._vote(why is None, device, group, why, 'mvae_comm_init')
        except Exception:
            if comm is not None:
                comm.destroy()
            raise
        return comm

    @staticmethod
    def _vote(ok, device, group, why, what):
        """All ranks agree (MIN) that a step worked everywhere; otherwise ALL raise."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            raise RuntimeError('%s failed on %s' % (what, ('this rank: %s' % why) if why else 'a peer'))

    @property
    def rccl_version(self):
        v = self._lib.mvae_comm_rccl_version()
        return '%d.%d.%d' % (v // 10000, (v // 100) % 100, v % 100) if v > 0 else None

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (what, rc, (self._lib.mvae_comm_last_error(self._h) or b'').decode()))

    def allreduce_async(self, t):
        """In-place sum of the contiguous fp32 tensor ``t`` over the ranks, ordered after the current stream's work,
        on the communicator's stream.  Returns the ticket for ``wait``."""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError('RcclComm.allreduce_async wants a contiguous fp32 GPU tensor')
        ticket = ctypes.c_int(-1)
        self._check(self._lib.mvae_comm_allreduce_async(self._h, ctypes.c_void_p(t.data_ptr()), t.numel(),
                                                        self._stream(), ctypes.byref(ticket)), 'mvae_comm_allreduce_async')
        return ticket.value

    def wait(self, ticket=-1):
        """The current stream waits for the collective behind ``ticket`` (-1: all issued so far)."""
        self._check(self._lib.mvae_comm_wait(self._h, int(ticket), self._stream()), 'mvae_comm_wait')

    def broadcast(self, t, root=0):
        if not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError('RcclComm.broadcast wants a contiguous GPU tensor')
        self._check(self._lib.mvae_comm_broadcast(self._h, ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(),
                                                  int(root), self._stream()), 'mvae_comm_broadcast')

    def self_check(self, device):
        """An all-reduce whose answer is known, eagerly and then from a captured hipGraph -- the two ways the engine
        issues collectives.  Raises if either gives a wrong sum."""
        expect = float(self.world * (self.world + 1) // 2)
        x = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device=device)
        self.wait(self.allreduce_async(x))
        torch.cuda.synchronize(device)
        if not bool((x == expect).all().item()):
            raise RuntimeError('eager all-reduce self-check: got %r, expected %r' % (x[0].item(), expect))
        y = torch.empty_like(x)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y.fill_(float(self.rank + 1))
                self.wait(self.allreduce_async(y))
            for _ in range(2):
                g.replay()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        if not bool((y == expect).all().item()):
            raise RuntimeError('captured all-reduce self-check: got %r, expected %r' % (y[0].item(), expect))

    def async_error(self):
        """Raise if a collective of this communicator has failed asynchronously (a peer died, a link error)."""
        self._check(self._lib.mvae_comm_async_error(self._h), 'mvae_comm_async_error')

    def synchronize(self, timeout_s=60.0):
        """The watchdog (``mvae_comm_synchronize``): block until the current stream's work -- after ``wait`` that
        includes the collectives -- has finished, for at most ``timeout_s``; raises RuntimeError when a peer failed or
        the budget ran out (a collective whose peer is gone never completes: ``torch.cuda.synchronize`` would hang
        for good).  After a raise the communicator must be abandoned (``abandon()``), not destroyed."""
        self._check(self._lib.mvae_comm_synchronize(self._h, self._stream(), int(timeout_s * 1000.0)),
                    'mvae_comm_synchronize')

    def abandon(self):
        """Forget the communicator without draining its stream (``destroy`` would wait for a collective that can
        never finish)."""
        self._h = None

    def destroy(self):
        if self._h is not None:
            self._lib.mvae_comm_destroy(self._h)
            self._h = None


class RcclBuckets(object):
    """``GradBuckets`` over the library's communicator: launch(k) / wait(k) are enqueue-only on the current
    stream (capturable)."""
    in_graph = True

    def __init__(self, flat_grad, ranges, comm):
        GradBuckets.__init__(self, flat_grad, ranges, group=None)      # range checks
        self.comm = comm
        self.pending = {}

    def launch(self, k):
        lo, hi = self.ranges[k]
        if k in self.pending:
            raise RuntimeError('bucket %d launched twice in one step' % k)
        self.pending[k] = self.comm.allreduce_async(self.flat[lo:hi])

    reset = GradBuckets.reset

    def wait(self, k=None):
        keys = sorted(self.pending) if k is None else [k]
        for key in keys:
            ticket = self.pending.pop(key, None)
            if ticket is not None:
                self.comm.wait(ticket)


class DataParallel(object):
    """Wrap a fused step engine: ``dp = DataParallel(model, engine)``.  Every
    ``engine.forward_backward`` then launches bucket k's all-reduce the moment the last weight gradient of
    its range has been enqueued, and ``dp.finish(optimizer)`` applies Adam bucket by bucket as the
    reductions land (``FusedAdam(..., grad_scale=dp.grad_scale)``): the update of bucket k overlaps the
    all-reduce of bucket k+1, and the last bucket -- the image encoder's first layers -- is small."""

    def __init__(self, model, engine, group=None, transport=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.world = dist.get_world_size(group)
        self.grad_scale = 1.0 / self.world
        arena = model.finalize()
        dist.broadcast(arena.flat, src=0, group=group)
        for b in model.buffers():
            dist.broadcast(b, src=0, group=group)
        ranges = bucket_ranges(model, arena)
        self.comm, self.transport = None, 'torch.distributed (%s)' % dist.get_backend(group)
        # an explicit ``transport=`` wins; MVAE_COMM only fills in for a caller that did not choose
        want = transport if transport is not None else os.environ.get('MVAE_COMM', 'rccl')
        if self.world > 1 and arena.flat.is_cuda and os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0') != '0':
            sys.stderr.write('[mvae parallel] HSA_ENABLE_IPC_MODE_LEGACY=%s: this driver only supports dmabuf IPC -- RCCL '
                             'across processes is expected to fail with hipIpcGetMemHandle: invalid argument; export '
                             'HSA_ENABLE_IPC_MODE_LEGACY=0 before the process first touches the GPU\n'
                             % os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])
        # the communicator needs torch.distributed only to carry its 128-byte id, so an EXPLICIT ``transport='rccl'``
        # also works over a gloo group (tests/test_comm_world2_gpu.py: two ranks on one GPU over a stand-in library);
        # the MVAE_COMM default only applies where torch.distributed itself runs on RCCL
        explicit = transport == 'rccl'
        if want == 'rccl' and (explicit or dist.get_backend(group) == 'nccl') and arena.flat.is_cuda:
            why = None
            try:
                self.comm = RcclComm.from_process_group(arena.flat.device, group)
                self.comm.self_check(arena.flat.device)
            except Exception as e:       # an error (not a hang) here is recoverable: the three-graph path needs nothing of it
                why = '%s: %s' % (type(e).__name__, e)
            # the choice of transport is COLLECTIVE: one rank on the fallback while its peers wait in the
            # communicator's all-reduce is a deadlock
            ok = torch.tensor([0 if why else 1], dtype=torch.int32, device=arena.flat.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 1:
                self.transport = 'mvae_comm (RCCL %s, collectives inside the step graph)' % self.comm.rccl_version
            else:
                sys.stderr.write('[mvae parallel] rank %d: C-ABI RCCL communicator unavailable here or on a peer (%s); '
                                 'all ranks fall back to torch.distributed all-reduces between three graphs\n'
                                 % (dist.get_rank(group), why or 'peer failed'))
                if self.comm is not None and why is None:
                    self.comm.destroy()
                self.comm = None
                self.transport += ' [fallback from mvae_comm]'
        if self.comm is not None:
            self.buckets = RcclBuckets(arena.grad, ranges, self.comm)
        else:
            self.buckets = GradBuckets(arena.grad, ranges, group=group)
        engine.configure_buckets(len(ranges))
        engine.on_bucket_ready = self.buckets.launch
        self.engine = engine

    @property
    def n_buckets(self):
        return len(self.buckets.ranges)

    @property
    def in_graph(self):
        """True: launch / wait / finish are enqueue-only and may be captured with the step."""
        return getattr(self.buckets, 'in_graph', False)

    def launch(self, k):
        """Start the all-reduce of bucket k (called by the engine between captured graphs)."""
        self.buckets.launch(k)

    def reset(self):
        self.buckets.reset()

    def wait(self, k=None):
        self.buckets.wait(k)

    def synchronize(self, timeout_s=60.0):
        """Host-side fence with a watchdog: like ``torch.cuda.synchronize()`` but raises RuntimeError when a peer has
        failed or the step has not finished within ``timeout_s`` (only the library's communicator can tell; the
        torch.distributed transport falls back to its own process-group timeout)."""
        if self.comm is not None:
            self.comm.synchronize(timeout_s)
        else:
            torch.cuda.synchronize()

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

"""Fused MVAE train step: the body of the reference's ``train(epoch)`` closure
(mnist/train.py:197-218, celeba/train.py:190-212, celeba19/train.py:257-308) as ONE batched
pass instead of three (celeba19: 20 + M) ``model()`` calls.

What the reference does per bimodal step: 2x image encoder, 2x label
encoder, 3x image decoder, 3x label decoder, 3x PoE + reparameterise, 3 ELBOs, one backward.
Here:

  * each encoder runs ONCE -- its output is identical in every call that includes the modality
    (BatchNorm sees the same batch, so the same statistics; the running statistics are advanced
    once per reference call).  CelebA's image encoder differs between its calls only in the
    Dropout(0.1) draw: the conv trunk + Linear(6400,512) + Swish run once, the masks are applied
    by a fan-out kernel and the final Linear runs on all the draws' rows;
  * all PoE / reparameterise / KL evaluations are one launch over the T terms;
  * each decoder runs once on the rows of all the terms that need it, BatchNorm statistics
    per term (``groups``), running statistics advanced in the reference's call order; decoder
    outputs the reference computes but never uses (mnist/train.py:208,211) are only evaluated
    where they have a side effect (BatchNorm running statistics -- SURVEY.md Appendix B-4);
  * the BCE / CE kernels emit the loss rows and d loss / d logits in one pass
    (d ELBO / d row is the constant lambda / B);
  * the backward is explicit (``layers.backward_tape``), weight gradients land in the
    gradient arena, and the whole step is a fixed launch sequence that ``capture()`` records
    into hipGraphs (``torch.cuda.CUDAGraph``) together with the optimizer.  Everything that
    changes from step to step (annealing factor, celeba19's sampled subsets) lives in small
    device tables refreshed from pinned host memory, so the graph never needs re-capture.

Results (per-term ELBOs, total, every gradient, BatchNorm running statistics) equal the
reference's multi-call step on the same noise; ``tests/test_engine_gpu.py`` checks that against
the oracle and the golden fixtures.
"""
import contextlib
import os

import torch

from . import kernels as K
from . import layers as L

KEEP = 0.9  # Dropout(p=0.1), celeba/model.py:91


def _snapshot(model, optimizer, counter):
    """Everything the warm-up iterations of a graph capture mutate."""
    bufs = [b.clone() for b in model.buffers()]
    pend = [(m, m._nbt_pending) for m in model.modules() if isinstance(m, L._BatchNormMixin)]
    opt = None
    if getattr(optimizer, '_arena', None) is not None:
        opt = (optimizer._m.clone(), optimizer._v.clone(), optimizer._step_dev.clone(), optimizer._host_step)
    return (model.arena.flat.clone(), bufs, pend, opt, counter.clone())


def _restore(model, optimizer, counter, snap):
    flat, bufs, pend, opt, ctr = snap
    model.arena.flat.copy_(flat)
    for b, s in zip(model.buffers(), bufs):
        b.copy_(s)
    for m, n in pend:
        m._nbt_pending = n
    counter.copy_(ctr)
    if getattr(optimizer, '_arena', None) is not None:
        if opt is None:       # bound during warm-up: back to a fresh optimizer
            optimizer._m.zero_(); optimizer._v.zero_(); optimizer._step_dev.zero_()
            optimizer._host_step = 0
        else:
            optimizer._m.copy_(opt[0]); optimizer._v.copy_(opt[1]); optimizer._step_dev.copy_(opt[2])
            optimizer._host_step = opt[3]


class StepTables(object):
    """The per-step device tables of a fused step (loss coefficients, PoE masks, BatchNorm update
    count) as ONE int32 block in HBM refreshed by ONE async copy from pinned host memory.

    The host may run several steps ahead of the GPU (graph replays are enqueue-only), so a pinned
    buffer must not be rewritten while an earlier step's copy may still be reading it: the host side is a
    ring of ``slots`` pinned blocks, each guarded by an event recorded right after its copy was queued;
    ``begin()`` waits on the slot's previous event (``slots`` steps back) before handing it out."""

    def __init__(self, n_words, device, slots=4):
        import numpy as np
        self.dev = torch.zeros(n_words, dtype=torch.int32, device=device)
        self._host, self._np, self._events = [], [], []
        for _ in range(slots):
            h = torch.zeros(n_words, dtype=torch.int32)
            if torch.cuda.is_available():
                h = h.pin_memory()
            self._host.append(h)
            self._np.append(h.numpy())
            self._events.append(None)
        self._i = 0
        self.defer = False
        self._pending = False
        self._np_float = [a.view(np.float32) for a in self._np]

    def ints(self, lo, n):
        return self.dev[lo:lo + n]

    def floats(self, lo, n):
        return self.dev[lo:lo + n].view(torch.float32)

    def begin(self):
        """(int32 view, float32 view) of the next pinned slot, safe to overwrite."""
        k = self._i % len(self._host)
        if self._events[k] is not None:
            self._events[k].synchronize()
        return self._np[k], self._np_float[k]

    def commit(self):
        if self.defer:
            self._pending = True        # replay(): the slot goes to the device with the batch, in ONE ingest launch
            return
        k = self._i % len(self._host)
        self.dev.copy_(self._host[k], non_blocking=True)
        self._mark(k)

    def _mark(self, k):
        if torch.cuda.is_available():
            if self._events[k] is None:
                self._events[k] = torch.cuda.Event()
            self._events[k].record()
        self._i += 1
        self._pending = False

    def pending_slot(self):
        """The pinned slot a deferred commit left to be sent (None: nothing pending)."""
        return self._host[self._i % len(self._host)] if self._pending else None

    def sent(self):
        """The pending slot was queued for transfer on the current stream (by an ingest launch)."""
        self._mark(self._i % len(self._host))


class _StepBase(object):
    """Launch plumbing shared by the fused steps: phase A (forward + decoder backward) and
    phase B (PoE + encoder backward) with a gradient-bucket hook after each, eager ``step``,
    and hipGraph capture -- one graph on a single GPU; with a communicator three graphs
    (A | B | optimizer) so the RCCL all-reduce of the decoder bucket, launched between A and B
    outside any capture, overlaps with phase B."""

    def _init_common(self, model, batch_size, seed):
        model.finalize()
        self.model = model
        self.B = int(batch_size)
        self.D = model.n_latents
        self.dev = next(model.parameters()).device
        self.seed = int(seed) or 1
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.on_bucket_ready = None     # parallel.py hooks gradient all-reduce launches here
        self.n_buckets = 2              # decoders | encoders; 3: the image encoder's first layers apart
        self._graphs = None
        self._comm = None
        self._bucket0_done = False
        # independent stacks (image vs label side) run as two branches; each kernel here fills
        # well under the 256 CUs, so the branches overlap instead of queueing
        n_streams = os.environ.get('MVAE_STREAMS', 'auto')
        wg_batched_auto = False
        if n_streams == 'auto':
            # conv stacks WITHOUT BatchNorm (FashionMNIST): the weight gradients are the step's longest launches (140-250 us
            # each) and have no BatchNorm backward between them to pace the chain -- on streams of their own (4) the step
            # is 2 % shorter than with the label branch's stream carrying them (2.082 -> 2.039 ms, x4 interleaved,
            # profiles/r06_sched_ab.txt); with BatchNorm (CelebA +1.8 %, CelebA-19 0) and for the MLP stacks (MNIST +25 %:
            # every fork is a cross-queue signal) two streams stay
            mods = list(model.modules())
            has_conv = any(isinstance(m, (L.Conv2d, L.ConvTranspose2d)) for m in mods)
            has_bn = any(isinstance(m, L._BatchNormMixin) for m in mods)
            # (MVAE_FUSE_ADAM=1, a non-default mode whose fused launches leave from the branch's own stream, keeps two)
            n_streams = 4 if (has_conv and not has_bn and os.environ.get('MVAE_FUSE_ADAM', '0') != '1') else 2
            # ... and everywhere else ONE further stream that takes the label decoder's weight-gradient BATCH (one launch for
            # its Linear layers + the conv closures) off the side stream, whose chain -- label decoder backward, then the label
            # encoder's -- no longer waits behind it: MNIST 0.2660 -> 0.2600 ms (-2.3 %, 5 of 5 interleaved rounds), CelebA
            # -0.4 %, CelebA-19 0 (profiles/r06_sched_ab.txt).  Single-GPU steps only: the data-parallel step sends its
            # decoder bucket from the side stream behind that batch (_launch_deferred, capture).
            if n_streams == 2 and os.environ.get('MVAE_FUSE_ADAM', '0') != '1' and os.environ.get('MVAE_PAIR_ENC', '0') != '1':
                n_streams, wg_batched_auto = 3, True
        self._streams_auto = os.environ.get('MVAE_STREAMS', 'auto') == 'auto' and int(n_streams) > 2
        n_streams = int(n_streams)
        self.side = torch.cuda.Stream(device=self.dev) if n_streams >= 2 else None
        # MVAE_STREAMS=3/4 (the default only for conv stacks without BatchNorm, above): each branch queues its weight-gradient
        # launches (layers.backward_tape ``deferred``) and runs them with ONE fork per backward
        # chain on a further stream, so only the data-gradient chains stay serial.  Measured on
        # MI355X: no gain (MNIST B=512 0.58-0.60 vs 0.56 ms/step, CelebA B=256 3.54-3.67 vs 3.58) --
        # with two branches in flight the kernels already fill the CUs; a fork per LAYER was 30 %
        # slower (every fork is a cross-queue signal).
        self.batch_wgrad = os.environ.get('MVAE_BATCH_WGRAD', '1') != '0'
        self.poe_draw = os.environ.get('MVAE_POE_DRAW', '1') != '0'      # eps drawn inside the PoE launch
        self.wgrad_on_side = os.environ.get('MVAE_WGRAD_SIDE', '1') != '0' and self.side is not None
        self.use_ingest = os.environ.get('MVAE_INGEST', '1') != '0'      # replay(): batch + tables in one launch
        # at a fork, issue the MAIN stream's continuation (the longer, image-side chain) BEFORE the side branch: the
        # graph keeps the successor that was recorded first on its producer's queue (0-3 us behind it) and reaches
        # the other over a cross-queue edge (9-12 us).  MVAE_MAIN_FIRST=0: side branch first (rounds 1-2).
        self.main_first = os.environ.get('MVAE_MAIN_FIRST', '1') != '0'
        # ... and at the DECODER fork.  Rounds 3-5 kept the side branch first there for two equal MLP decoders (MNIST: the
        # side branch also carries both decoders' weight gradients, 'auto'); on round 6's kernels main-first is 1.3 %
        # faster on MNIST as well (0.2699 -> 0.2663 ms, 6 of 6 interleaved rounds, profiles/r06_sched_ab.txt).
        # MVAE_MAIN_FIRST_DEC=0|auto: side first always | unless the image decoder is a conv stack.
        self.main_first_dec = os.environ.get('MVAE_MAIN_FIRST_DEC', '1')
        # a decoder that ends in a plain Linear: that launch also evaluates the reconstruction term (the logits never
        # reach memory); MVAE_LOSS_FOLD=0: Linear, then the loss kernel.  'image' / 'label': only that decoder.
        fold = os.environ.get('MVAE_LOSS_FOLD', '1')
        self.fold_image = fold in ('1', 'image')
        self.fold_label = fold in ('1', 'label')
        # one-graph data-parallel step: buckets 0 and 1 go out from the SIDE stream (MVAE_DP_SIDE_LAUNCH=0: from the
        # main stream behind a full join)
        self.dp_side_launch = os.environ.get('MVAE_DP_SIDE_LAUNCH', '1') != '0'
        self._bucket1_done = False
        self._draw_in_poe = False
        self.batch_repack = os.environ.get('MVAE_BATCH_REPACK', '1') != '0'
        self._conv_mods = [m for m in model.modules() if isinstance(m, (L.Conv2d, L.ConvTranspose2d))]
        self.wg_main = torch.cuda.Stream(device=self.dev) if n_streams >= 3 else None
        # the branches keep their batched weight-gradient launches and the label decoder's batch runs on the further stream
        self.wg_batched = os.environ.get('MVAE_WG_BATCHED', '1' if wg_batched_auto else '0') == '1'
        self.wg_side = torch.cuda.Stream(device=self.dev) if n_streams >= 4 else self.wg_main
        self._wg_pending = []
        self._forked = False
        # single-GPU captured step: the Linear weight-gradient batches apply Adam to their own outputs (see
        # _single_gpu_step); set per model family by the subclass, None while no such step is being issued
        self.fuse_adam = False
        self._fusion = None

    @contextlib.contextmanager
    def _branch(self, after=None):
        """Run the body on the side stream, ordered after everything launched so far -- or, with ``after`` (an event
        from ``_fork_point``), after that point only: the main stream's own continuation can then be issued FIRST.
        Tensors the body allocates must stay referenced until the next fork (``_carry``)."""
        if self.side is None:
            yield
            return
        if after is None:
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
        else:
            self.side.wait_event(after)
        self._forked = True
        with torch.cuda.stream(self.side):
            yield

    def _fork_point(self):
        """An event at the current end of the main stream (None without a side stream or with MVAE_MAIN_FIRST=0)."""
        if self.side is None or not self.main_first:
            return None
        ev = torch.cuda.Event()
        ev.record()
        self._carry.setdefault('fork_events', []).append(ev)
        return ev

    def _deferred(self):
        """The list a backward chain queues its weight-gradient launches in (None: launch inline).  Default:
        a ``layers.WgradBatch`` -- the chain's Linear weight gradients leave as ONE launch when the chain is
        done (MVAE_BATCH_WGRAD=0: inline, one launch per layer)."""
        if self.wg_main is not None and not self.wg_batched:
            return []
        return L.WgradBatch(adam=self._fusion) if self.batch_wgrad else None

    def _launch_deferred(self, fns, stream):
        """Run the queued weight-gradient launches on ``stream``, ordered after everything the
        CURRENT stream has launched.  Joined by ``_join_wgrad`` -- directly into the origin stream:
        hipGraph capture (ROCm 7.0) crashes in EndCapture when a fork of a fork joins back into
        its parent (tools/graph_fork_probe.py: 'nested' vs 'nested_join_main').  A ``WgradBatch`` is
        flushed on the current stream instead."""
        if isinstance(fns, L.WgradBatch) and not (self.wg_batched and stream is not None and len(fns)
                                                 and self._comm is None and self.on_bucket_ready is None):
            fns.flush()
            return
        if not fns:
            return
        stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(stream):
            if isinstance(fns, L.WgradBatch):
                fns.flush()                 # MVAE_WG_BATCHED=1: the batch (one Linear launch + the conv closures) on the weight-gradient stream
            else:
                for fn in fns:
                    fn()
        self._carry.setdefault('deferred', []).append(fns)     # the closures own the gradients
        if stream not in self._wg_pending:
            self._wg_pending.append(stream)

    def _join_wgrad(self):
        main = torch.cuda.current_stream(self.dev)
        for s in self._wg_pending:
            main.wait_stream(s)
        self._wg_pending = []

    def _join(self):
        if self._forked:
            main = torch.cuda.current_stream(self.dev)
            main.wait_stream(self.side)
            self._forked = False

    def configure_buckets(self, n):
        """Gradient buckets of the data-parallel exchange (parallel.bucket_ranges): 2 = decoders | encoders,
        3 = decoders | encoders without the image encoder's first layers | those layers (arena tail)."""
        if n not in (1, 2, 3):
            raise ValueError('1, 2 or 3 gradient buckets')
        if n == 3 and self.model.arena.tail_range is None:
            raise ValueError('three buckets need an arena tail')
        self.n_buckets = n

    def _phases_b(self):
        """The launch groups of phase B, one per encoder bucket."""
        if self.n_buckets == 3:
            return [lambda: self._phase_b('upper'), lambda: self._phase_b('lower')]
        return [lambda: self._phase_b('all')]

    # subclasses: _phase_a(image, label), _phase_b(part), set_coefficients(beta), draw_noise()
    def forward_backward(self, image, label):
        """Launch the whole forward + backward.  Gradients go to ``p.grad`` (the arena); returns
        the device tensor ``elbo[T+1]`` = per-term ELBOs (engine order) and their sum."""
        self._step_begin()
        try:
            self._phase_a(image, label)
            if self.on_bucket_ready is not None and self.n_buckets > 1:
                self.on_bucket_ready(0)      # decoder gradients are final
            for k, part in enumerate(self._phases_b()):
                part()
                if self.on_bucket_ready is not None:
                    self.on_bucket_ready(k + 1 if self.n_buckets > 1 else 0)   # this group of encoder gradients is final
        finally:
            self._step_end()         # also when a launch raised: a later forward must not read this step's weight copies
        return self.elbo

    def _step_begin(self):
        """Ahead of the step: the repacked weight copies of all dgrad-form conv launches in ONE launch (the
        launches otherwise each make their own, a 5-us kernel in front of every one of them on the chain)."""
        if self.batch_repack:
            L.repack_weights(self._conv_mods)

    def _step_end(self):
        if self.batch_repack:
            L.repack_done(self._conv_mods)      # the optimizer changes the weights next

    def step(self, image, label, annealing_factor, noise=None):
        """One eager step (no optimizer): zero_grad -> forward/backward.  Returns elbo[T+1]."""
        self.model.zero_grad(set_to_none=True)
        self.set_coefficients(annealing_factor)
        if noise is not None:
            self.set_noise(noise)
        else:
            self.draw_noise()
        return self.forward_backward(image, label)

    # ------------------------------------------------------------------ hipGraph capture
    def capture(self, optimizer, image_shape, label_example, warmup=3, comm=None):
        """Record zero_grad + forward/backward + optimizer.step() into hipGraph(s).  After this,
        ``replay(image, label, beta)`` copies the batch into static buffers, refreshes the
        device tables and launches the graph(s): no per-kernel host work.  ``comm`` is a
        ``parallel.DataParallel`` (or anything with launch(k) / wait())."""
        if comm is not None and getattr(self, '_streams_auto', False) and self.wg_main is not None:
            # the data-parallel step sends its buckets from the side stream as the gradients become final: the weight
            # gradients stay on it (FashionMNIST at world 1 through mvae_comm: 2.17 ms with them on their own streams, 2.08 without)
            self.wg_main = self.wg_side = None

        dev = self.dev
        self.static_image = torch.zeros((self.B,) + tuple(image_shape), dtype=torch.float32, device=dev)
        self.static_label = torch.zeros_like(label_example, device=dev)
        self.set_coefficients(1.0)
        self._comm = comm
        hook, self.on_bucket_ready = self.on_bucket_ready, None
        snap = _snapshot(self.model, optimizer, self.counter)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        bns = [m for m in self.model.modules() if isinstance(m, L._BatchNormMixin)]
        in_graph = comm is not None and getattr(comm, 'in_graph', False)
        with torch.cuda.stream(side):
            for it in range(warmup):
                before = [m._nbt_pending for m in bns]
                if in_graph:
                    self._dp_step(optimizer)       # with the real collectives: RCCL sets its connections up OUTSIDE the capture
                else:
                    self._single_gpu_step(optimizer)
                # graph replays skip the host code that counts BatchNorm calls: remember the
                # per-step increments of num_batches_tracked and re-apply them in replay()
                self._bn_inc = [(m, m._nbt_pending - b) for m, b in zip(bns, before)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        if comm is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._single_gpu_step(optimizer)
            self._graphs = (g,)
        elif in_graph:
            # data parallel over the library's communicator (mvae_comm_*): launch / wait are stream + event operations
            # and one RCCL enqueue -- the WHOLE step is one graph: forward, backward, the bucket all-reduces on the
            # communicator's stream (forked from / joined into this one), per-bucket Adam, the counter launch
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._dp_step(optimizer)
            self._graphs = (g,)
            self._optimizer = optimizer
        else:
            # data parallel: one graph per gradient bucket (A = forward + decoder backward, then one or two
            # groups of encoder backward); the bucket all-reduces are issued between the replays, outside any
            # capture, and Adam runs per bucket as they land (parallel.DataParallel.finish) -- eager launches,
            # one kernel per bucket
            pool = torch.cuda.graph_pool_handle()
            graphs = [torch.cuda.CUDAGraph()]
            try:
                with torch.cuda.graph(graphs[0], pool=pool):
                    self._body_a()
                for part in self._phases_b():
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool):
                        part()
                    graphs.append(g)
            finally:
                self._step_end()
            self._graphs = tuple(graphs)
            self._optimizer = optimizer
        _restore(self.model, optimizer, self.counter, snap)   # warm-up steps must leave no trace
        torch.cuda.synchronize(dev)
        self.on_bucket_ready = hook
        return self._graphs

    def _single_gpu_step(self, optimizer):
        """The captured single-GPU step.  With the fused optimizer the step counter is advanced on the side stream behind
        the label / attribute encoders' forward -- the shorter encoder branch; a one-thread launch, nothing waits for it -- and the update at the end of the chain
        is ONE launch at t = counter (MVAE_EARLY_COUNTER=0: update + counter launch at the end).
        ``fuse_adam`` (MVAE_FUSE_ADAM=1, off -- measured slower, see BimodalStep): the weight-gradient batches update their
        own parameters (layers.WgradBatch(adam=...)), the counter launch also leaves the step's bias corrections for them,
        and what remains for the end of the chain is Adam over the ranges no batch covered -- on MNIST none."""
        early = (os.environ.get('MVAE_EARLY_COUNTER', '1') != '0' and self.side is not None
                 and hasattr(optimizer, 'step_counted'))
        self._adam_counter = optimizer.step_counter() if early else None
        self._optimizer_early = optimizer if early else None
        fuse = (early and self.fuse_adam and self.batch_wgrad and self.wg_main is None
                and hasattr(optimizer, 'fusion') and optimizer.grad_scale == 1.0)
        self._fusion = optimizer.fusion() if fuse else None
        if self._fusion is not None:
            self._fusion.begin()
            self._adam_prepare = lambda: optimizer.prepare_counted(self._fusion)
        # MVAE_SPLIT_ADAM=1: the decoders' parameters (the front of the arena) are updated on the MAIN stream as soon as
        # their weight-gradient batches -- on the side stream -- are final, beside the encoders' backward; only the encoders'
        # range is left for the end of the chain (see BimodalStep._phase_b)
        self._adam_split = None
        if (early and self._fusion is None and os.environ.get('MVAE_SPLIT_ADAM', '0') == '1'
                and hasattr(optimizer, 'step_counted_range') and getattr(self, 'supports_split_adam', False)):
            from .parallel import bucket_ranges
            ranges = bucket_ranges(self.model, self.model.arena)
            if len(ranges) >= 2:
                self._adam_split = (optimizer, ranges[0][1], self.model.arena.numel)
        try:
            self._body_a()
            self._phase_b('all')
            if self._adam_split is not None and self._carry.get('adam_dec_done'):
                optimizer.step_counted_range(self._adam_split[1], self._adam_split[2], last=True)
            elif early:
                optimizer.step_counted(self._fusion) if self._fusion is not None else optimizer.step_counted()
            else:
                optimizer.step()
        finally:
            self._step_end()
            self._adam_counter = None
            self._fusion = None
            self._adam_split = None

    def _dp_step(self, optimizer):
        """One data-parallel step as a plain launch sequence (capturable when the communicator is): bucket k's
        all-reduce goes out the moment its gradients are final and runs behind the remaining backward; Adam follows
        bucket by bucket (parallel.DataParallel.finish)."""
        comm = self._comm
        self._adam_counter = None
        self._bucket0_done = self._bucket1_done = False
        if hasattr(comm, 'reset'):
            comm.reset()            # tickets of a step that raised half-way must not poison this one
        try:
            self._body_a()
            if self.n_buckets > 1 and not self._bucket0_done:      # else: phase A sent it from the side stream
                comm.launch(0)
            for k, part in enumerate(self._phases_b()):
                part()
                bucket = k + 1 if self.n_buckets > 1 else 0
                if not (bucket == 1 and self._bucket1_done):        # else: phase B's first half sent it from the side stream
                    comm.launch(bucket)
        finally:
            self._step_end()
        self._join()            # every stream of the step is back on this one before the optimizer (and the end of a capture)
        comm.finish(optimizer)

    def _poe_forward(self, mus, lvs, mu, lv, z, kl):
        """PoE + reparameterise + KL for all the step's terms; eps either sits in ``self.noise`` (set_noise /
        philox_fill) or is drawn by the launch itself into ``self.noise`` (same Philox stream, offset 0)."""
        if self._draw_in_poe:
            K.poe_fwd_draw(mus, lvs, self.masks_dev, self.noise, self.seed, self.counter, 0, mu, lv, z, kl,
                           self.model.POE_VARIANT)
        else:
            K.poe_fwd(mus, lvs, self.masks_dev, self.noise, mu, lv, z, kl, self.model.POE_VARIANT)

    def _side_launch_ok(self):
        """The one-graph data-parallel step may issue a bucket's all-reduce from the side stream: the communicator's
        launch is a stream operation (parallel.RcclBuckets), so the main stream need not join the side stream -- and
        wait for its weight gradients -- just to start a collective."""
        return (self.dp_side_launch and self.side is not None and self._comm is not None
                and getattr(self._comm, 'in_graph', False) and self.on_bucket_ready is None)

    def _upper_done(self):
        """End of phase B's first half with three buckets: bucket 1 (the encoders without the image encoder's first
        layers) is final once BOTH streams are here.  Old form: join, then the caller launches from the main stream --
        which made the main stream wait for the side stream's weight gradients before the conv trunk's backward.
        Now the side stream waits for the main stream's event and launches; the main stream goes straight on."""
        if self._side_launch_ok() and self.n_buckets == 3:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self._comm.launch(1)
            self._carry.setdefault('events', ())
            self._carry['events'] += (ev,)
            self._bucket1_done = True
            self._forked = True
        else:
            self._join()

    def _early_counter(self):
        """Called at the end of the side stream's encoder forward (the shorter of the two encoder branches)."""
        if getattr(self, '_adam_counter', None) is not None:
            if self._fusion is not None:
                self._adam_prepare()        # counter + the bias corrections the fused weight-gradient launches read
            elif hasattr(self._optimizer_early, 'prepare_plain'):
                self._optimizer_early.prepare_plain()   # counter + the step's bias corrections for the update at the end
            else:
                K.counter_add(self._adam_counter, 1)

    def _body_a(self):
        self.model.zero_grad(set_to_none=True)
        self._step_begin()
        self.draw_noise()
        self._phase_a(self.static_image, self.static_label)

    def _bump_bn_counters(self):
        for m, inc in self._bn_inc:
            m._nbt_pending += inc

    def replay(self, image, label, annealing_factor):
        self._bump_bn_counters()
        if self.use_ingest and K.ingest_ok(image, self.static_image, label, self.static_label):
            # batch + per-step tables in ONE launch (the table slot is pinned host memory the kernel reads itself)
            self.tables.defer = True
            try:
                self.set_coefficients(annealing_factor)
            finally:
                self.tables.defer = False
            slot = self.tables.pending_slot()
            K.ingest(image, self.static_image, label, self.static_label, slot, self.tables.dev)
            self.tables.sent()
        else:
            self.static_image.copy_(image, non_blocking=True)
            self.static_label.copy_(label, non_blocking=True)
            self.set_coefficients(annealing_factor)
        if self._comm is None or getattr(self._comm, 'in_graph', False):
            self._graphs[0].replay()         # single GPU, or data parallel with the collectives inside the graph
        else:
            # graph 0 = phase A (decoder gradients final), graphs 1.. = the encoder groups: bucket k's all-reduce
            # runs behind the next graph / the Adam launches.  With ONE bucket (a model without encoder ranges)
            # nothing is reduced after phase A -- the whole arena goes out behind the last graph, exactly as
            # forward_backward() does it eagerly
            for k, g in enumerate(self._graphs):
                g.replay()
                if self.n_buckets > 1:
                    self._comm.launch(k)
                elif k == len(self._graphs) - 1:
                    self._comm.launch(0)
            self._comm.finish(self._optimizer)
        return self.elbo


class BimodalStep(_StepBase):
    supports_split_adam = True
    """Fused step for ``mnist`` / ``fashionmnist`` / ``celeba`` MVAEs."""

    def __init__(self, model, batch_size, lambda_image=1.0, lambda_label=1.0, seed=0):
        self._init_common(model, batch_size, seed)
        self.lambda_image = float(lambda_image)
        self.lambda_label = float(lambda_label)
        self.has_dropout = L.n_dropout(model.image_encoder.plan()) > 0
        self.has_bn = bool(model.HAS_BN)
        B, dev = self.B, self.dev
        if self.has_dropout:
            # experts: image(call 1), image(call 2), label; terms in the reference's call order
            self.term_masks = [0b101, 0b010, 0b100]
            self.ref_order = [0, 1, 2]                      # engine term -> reference term
            self.trunk = L.compile_plan(model.image_encoder.trunk_modules())
            self.head = L.compile_plan(model.image_encoder.head_modules())
        else:
            # no BatchNorm side effects to order: [image, joint, label] makes both decoders'
            # rows contiguous
            self.term_masks = [0b01, 0b11, 0b10]
            self.ref_order = [1, 0, 2]
        self.T = 3
        if self.has_bn:
            self.img_terms, self.lbl_terms = (0, 2), (0, 3)   # (first term, count) with loss
            self.img_has_loss = [1, 1, 0]
            self.lbl_has_loss = [1, 0, 1]
        else:
            self.img_terms, self.lbl_terms = (0, 2), (1, 2)
            self.img_has_loss = [1, 1, 0]
            self.lbl_has_loss = [0, 1, 1]
        self.masks_dev = torch.tensor(self.term_masks, dtype=torch.int32, device=dev)
        # MVAE_PAIR=1 (tuning aid, off): run the decoders' shared leading layers as paired launches.  Measured
        # on MI355X: 17 % fewer graph nodes but no gain (MNIST B=512 0.508 vs 0.493 ms/step) -- the two
        # decoders already overlap on two streams and a paired launch serialises them.
        self.pair_dec = self._pairable_decoder_layers() if os.environ.get('MVAE_PAIR', '0') == '1' else 0
        # MVAE_PAIR_ENC=1 (tuning aid, off): the two ENCODERS' trailing layers of equal shape (MNIST: 512 -> 512 and the
        # 512 -> 2D heads, mnist/model.py:76-78,117-119) as G = 2 launches on ONE stream -- the encoder phases then have no
        # fork and no join, 35 -> 30 launches per step.  Measured on MI355X (profiles/r04_xcd_bn_pair_ab.txt, three
        # interleaved pairs): 0.2996 / 0.2968 / 0.2987 ms paired against 0.2967 / 0.2949 / 0.2960 on two branches --
        # 1 % SLOWER.  The side branch is real parallel work (its kernels run beside the image side's on idle CUs); a
        # G = 2 launch takes as long as the two it replaces and the fork/join edge it removes is cheaper than that.
        self.pair_enc = self._pairable_encoder_layers() if os.environ.get('MVAE_PAIR_ENC', '0') == '1' else 0
        # each decoder's backward runs to the latent on its own stream into its own buffer (poe_bwd_split adds
        # them); MVAE_SPLIT_DZ=0: one cleared dz both first layers accumulate into after the join
        self.split_dz = os.environ.get('MVAE_SPLIT_DZ', '1') != '0'
        # MVAE_FUSE_ADAM=1 (tuning aid, off): the Linear weight-gradient batches run optimizer.step() on their own
        # outputs (_single_gpu_step, mvae_linear_wgrad_batched_adam) and -- on the all-Linear MNIST model -- nothing is
        # left for a launch at the end of the chain.  Bit-identical parameters (tests), but measured on MI355X
        # (profiles/r04_fuse_adam_ab.txt, three interleaved rounds): MNIST B=512 0.3042 / 0.3055 / 0.3038 ms fused
        # against 0.2988 / 0.2990 / 0.2988 with the arena-wide launch -- 1.9 % SLOWER (3 % before the parameter and
        # moments were prefetched ahead of the reduction); forced on the conv models +0.9 % / +1.3 %.  The arena-wide
        # launch streams its 28 B / parameter at 5 TB/s in 15 us; the same traffic inside the batches -- which sit in
        # FRONT of the label encoder's backward on the side stream -- lengthens them by more than the launch it removes.
        # Sound only where nothing reads a layer's weights after its weight-gradient batch went out: each decoder's
        # backward must run to the latent inside its own chain (split_dz) and the decoders may not be paired.
        # MVAE_COUNTER_RIDES=decoder: the step-counter launch on the label DECODER's branch instead of behind the label
        # encoder's forward -- same A/B: 0.2988 vs 0.2980 ms, nothing.
        self.counter_rides = os.environ.get('MVAE_COUNTER_RIDES', 'encoder')
        self.fuse_adam = (os.environ.get('MVAE_FUSE_ADAM', '0') == '1' and self.split_dz and not self.pair_dec)
        # per-term loss coefficients lambda/B, beta/B: pinned host mirror -> device, so a captured
        # graph sees new annealing factors without re-capture
        self.tables = StepTables(3 * self.T, dev)
        self.coef = self.tables.floats(0, 3 * self.T).reshape(3, self.T)
        self.noise = torch.empty(self.T, B, self.D, dtype=torch.float32, device=dev)
        self.drop_masks = torch.empty(2, B, 512, dtype=torch.float32, device=dev) if self.has_dropout else None
        self.elbo = torch.zeros(self.T + 1, dtype=torch.float32, device=dev)

    def _pairable_decoder_layers(self):
        """How many leading layers of the image and the label decoder are the same plain
        Linear + Swish shape on the same number of rows (MNIST: 3, mnist/model.py:96-104,137-145;
        FashionMNIST: 1): those run as ONE launch for both decoders (kernels.linear_*_pair)."""
        if self.has_bn or self.img_terms[1] != self.lbl_terms[1]:
            return 0
        pi, pl = self.model.image_decoder.plan(), self.model.label_decoder.plan()
        n = 0
        for a, b in zip(pi, pl):
            if not (a.kind == 'lin' and b.kind == 'lin' and a.act and b.act and a.drop == 0 and b.drop == 0
                    and a.mod.weight.shape == b.mod.weight.shape
                    and (a.mod.bias is None) == (b.mod.bias is None)):
                break
            n += 1
        return n if n < min(len(pi), len(pl)) else 0      # both stacks need a tail (their output layers)

    def _pairable_encoder_layers(self):
        """How many TRAILING layers of the image and the label encoder are the same Linear shape (MNIST: 2 --
        fc2 and the fc31 | fc32 head pair) with nothing but ``View + Linear + Swish`` (image) and ``Embedding + Swish``
        (label) in front of them.  0: the encoders run as two branches."""
        if self.has_bn or self.has_dropout or self.side is None or not self.batch_wgrad or self.wg_main is not None:
            return 0
        pi, pl = self.model.image_encoder.plan(), self.model.label_encoder.plan()
        n = 0
        while n < min(len(pi), len(pl)):
            a, b = pi[-1 - n], pl[-1 - n]
            if not (a.kind == b.kind and a.kind in ('lin', 'lin2') and a.act == b.act and a.drop == 0 and b.drop == 0):
                break
            wa, ba = L._lin_weights(a)
            wb, bb = L._lin_weights(b)
            if wa.shape != wb.shape or (ba is None) != (bb is None):
                break
            n += 1
        head_i = [op for op in pi[:len(pi) - n] if op.kind != 'view']
        head_l = pl[:len(pl) - n]
        ok = (n >= 1 and len(head_i) == 1 and head_i[0].kind == 'lin' and head_i[0].act and head_i[0].drop == 0
              and len(head_l) == 1 and head_l[0].kind == 'emb')
        return n if ok else 0

    def _pair_enc_now(self):
        return self.pair_enc and self.n_buckets != 3

    def _encoders_paired_fwd(self, image, lbl_in):
        """Both encoders on this stream: Embedding, the image side's first Linear, then the shared-shape layers as
        paired launches.  Returns (image heads, label heads) [B, 2D]."""
        m, B, P, dev, c = self.model, self.B, self.pair_enc, self.dev, self._carry
        pi, pl = m.image_encoder.plan(), m.label_encoder.plan()
        ni, nl = len(pi) - P, len(pl) - P
        h_l, tape_l = L.forward_tape(pl[:nl], lbl_in)
        h_i, tape_i = L.forward_tape(pi[:ni], image)
        h, layers = (h_i, h_l), []
        for j in range(P):
            wbs = [L._lin_weights(pi[ni + j]), L._lin_weights(pl[nl + j])]
            w = (wbs[0][0].detach(), wbs[1][0].detach())
            b = (None, None) if wbs[0][1] is None else (wbs[0][1].detach(), wbs[1][1].detach())
            N = w[0].shape[0]
            pre = torch.empty(2, B, N, dtype=torch.float32, device=dev)
            if pi[ni + j].act:
                act = torch.empty(2, B, N, dtype=torch.float32, device=dev)
                K.linear_fwd_pair(h, w, b, (pre[0], pre[1]), (act[0], act[1]))
                layers.append((h, pre, w))
                h = (act[0], act[1])
            else:
                K.linear_fwd_pair(h, w, b, (pre[0], pre[1]), (None, None))
                layers.append((h, None, w))
                h = (pre[0], pre[1])
        c['enc_pair'] = (layers, tape_i, tape_l, ni, nl)
        return h

    def _encoders_paired_bwd(self, g_img, g_lbl):
        """Backward of ``_encoders_paired_fwd``: paired data gradients, then ONE weight-gradient batch for both
        encoders and the Embedding gradient -- no fork, no join."""
        m, B, P, dev, c = self.model, self.B, self.pair_enc, self.dev, self._carry
        pi, pl = m.image_encoder.plan(), m.label_encoder.plan()
        layers, tape_i, tape_l, ni, nl = c['enc_pair']
        wb = L.WgradBatch(adam=self._fusion)
        g = (g_img, g_lbl)
        keep = [g]
        first_i = [k for k, op in enumerate(pi[:ni]) if op.kind == 'lin'][0]
        for j in range(P - 1, -1, -1):
            x, _, w = layers[j]
            wb.add_linear(pi[ni + j], g[0], x[0])
            wb.add_linear(pl[nl + j], g[1], x[1])
            K_in = w[0].shape[1]
            dx = torch.empty(2, B, K_in, dtype=torch.float32, device=dev)
            if j > 0:
                pre_in = layers[j - 1][1]
                pin = (pre_in[0], pre_in[1])
            else:
                # the image side's producer is an activated Linear (its Swish' rides this launch), the label side's is
                # the Embedding, whose own backward applies Swish'.  A pair shares its epilogue: the label side reads a
                # pre-activation whose Swish' is exactly 1.0f (sigmoid(1e4) == 1.0f: s * (1 + x * (1 - s)) == 1)
                if getattr(self, '_unit_pre', None) is None or self._unit_pre.shape != (B, K_in):
                    self._unit_pre = torch.full((B, K_in), 1e4, dtype=torch.float32, device=dev)
                pin = (tape_i[first_i][1], self._unit_pre)
            K.linear_dgrad_pair(g, w, (dx[0], dx[1]), pin)
            g = (dx[0], dx[1])
            keep.append(dx)
        wb.add_linear(pi[first_i], g[0], tape_i[first_i][0])         # first layer: no data gradient
        wb.flush()
        emb = pl[0].mod
        dw, acc = L.grad_target(emb.weight)
        K.embedding_swish_bwd(tape_l[0][0], emb.weight.detach(), g[1].contiguous(), dw, accumulate=acc)
        c['enc_pair_keep'] = (keep, wb)

    def _decoders_paired(self, z, image, label, lbl_in):
        """Decoder forward, reconstruction terms + gradients, decoder backward with the first
        ``pair_dec`` layers of both decoders as paired launches.  Returns what the unpaired path
        computes: (g_img, g_lbl) = gradients at the first layers' outputs, the loss rows, keep-alive."""
        m, B, D, P = self.model, self.B, self.D, self.pair_dec
        dev = self.dev
        (i0, ni), (l0, nl) = self.img_terms, self.lbl_terms
        R = ni * B
        plans = (m.image_decoder.plan(), m.label_decoder.plan())
        h = (z[i0:i0 + ni].reshape(R, D), z[l0:l0 + nl].reshape(R, D))
        layers = []                                   # per paired layer: (inputs, pre[2,R,N], weights, biases)
        for j in range(P):
            wb = [L._lin_weights(pl[j]) for pl in plans]
            w = (wb[0][0].detach(), wb[1][0].detach())
            b = (None, None) if wb[0][1] is None else (wb[0][1].detach(), wb[1][1].detach())
            N = w[0].shape[0]
            pre = torch.empty(2, R, N, dtype=torch.float32, device=dev)
            act = torch.empty(2, R, N, dtype=torch.float32, device=dev)
            K.linear_fwd_pair(h, w, b, (pre[0], pre[1]), (act[0], act[1]))
            layers.append((h, pre, w))
            h = (act[0], act[1])
        N_last = h[0].shape[1]
        gbuf = torch.empty(2, R, N_last, dtype=torch.float32, device=dev)      # d loss / d act of layer P-1
        # ---- tails: label on the side stream, image here
        with self._branch():
            logits_lbl, tape_tl = L.forward_tape(plans[1][P:], h[1], groups=nl)
            rows_lbl = torch.empty(nl * B, dtype=torch.float32, device=dev)
            dlog_lbl = torch.empty_like(logits_lbl)
            if m.LABEL_KIND == 'class':
                K.ce_fwd(logits_lbl, label, rows_lbl, drow=self.coef[1, l0:l0 + nl], dlogits=dlog_lbl,
                         rows_per_group=B, label_rows=B)
            else:
                K.bce_rowsum_fwd(logits_lbl, lbl_in, rows_lbl, drow=self.coef[1, l0:l0 + nl],
                                 dlogits=dlog_lbl, rows_per_group=B, target_rows=B)
            L.backward_tape(plans[1][P:], tape_tl, dlog_lbl, need_input_grad=True, groups=nl,
                            input_grad_out=gbuf[1], input_grad_accumulate=False)
        logits_img, tape_ti = L.forward_tape(plans[0][P:], h[0], groups=ni)
        npix = logits_img[0].numel()
        li = logits_img.reshape(ni * B, npix)
        rows_img = torch.empty(ni * B, dtype=torch.float32, device=dev)
        dlog_img = torch.empty_like(li)
        K.bce_rowsum_fwd(li, image.reshape(B, npix), rows_img, drow=self.coef[0, i0:i0 + ni], dlogits=dlog_img,
                         rows_per_group=B, target_rows=B)
        L.backward_tape(plans[0][P:], tape_ti, dlog_img.reshape(logits_img.shape), need_input_grad=True,
                        groups=ni, input_grad_out=gbuf[0], input_grad_accumulate=False)
        self._join()
        # ---- paired backward: Swish' of the last paired layer, then weight + data gradients per layer
        g = torch.empty_like(gbuf)
        K.swish_bwd(gbuf, layers[P - 1][1], g)
        keep = [layers, gbuf, logits_lbl, tape_tl, dlog_lbl, logits_img, tape_ti, dlog_img, g]
        for j in range(P - 1, -1, -1):
            x, _, w = layers[j]
            targets = []
            for pl in plans:
                mod = pl[j].mod
                dw, acc = L.grad_target(mod.weight)
                db, acc_b = (None, acc) if mod.bias is None else L.grad_target(mod.bias)
                if acc != acc_b:
                    raise RuntimeError('Linear weight/bias gradients out of sync')
                targets.append((dw, db, acc))
            if targets[0][2] != targets[1][2]:
                raise RuntimeError('paired gradients out of sync')
            K.linear_wgrad_pair((g[0], g[1]), x, (targets[0][0], targets[1][0]), (targets[0][1], targets[1][1]),
                                accumulate=targets[0][2])
            if j > 0:
                pre_in = layers[j - 1][1]
                dx = torch.empty(2, R, w[0].shape[1], dtype=torch.float32, device=dev)
                K.linear_dgrad_pair((g[0], g[1]), w, (dx[0], dx[1]), (pre_in[0], pre_in[1]))
                keep.append(g)
                g = dx
        keep.append(g)
        return g[0], g[1], rows_img, rows_lbl, keep

    # ------------------------------------------------------------------ reconstruction term inside the last Linear
    def _fold_plan(self, plan, kind):
        """Can the stack's last launch carry its reconstruction term (``kind``: 'bce' or 'class')?"""
        op = plan[-1]
        if not (op.kind == 'lin' and not op.act and op.drop == 0):
            return False
        return kind == 'bce' or op.mod.weight.shape[0] <= 32

    def _decode_with_loss(self, plan, zin, groups, kind, target, drow, which):
        """Decoder forward with the reconstruction term folded into the last Linear's launch.
        Returns (d loss / d logits, tape, loss rows, rows per group of the loss rows)."""
        B = self.B
        N = plan[-1].mod.weight.shape[0]
        if kind == 'bce':
            nparts = K.bce_partials(N)
            rows = torch.empty(groups * B * nparts, dtype=torch.float32, device=self.dev)
            tgt = target.reshape(B, N)

            def fold(x, w, b, out, logits=None, rows=rows):
                K.linear_bce_fwd(x, w, b, tgt, drow, out, rows, B, B, logits=logits)
            rpg = B * nparts
        else:
            rows = torch.empty(groups * B, dtype=torch.float32, device=self.dev)

            def fold(x, w, b, out, logits=None, rows=rows):
                K.linear_ce_fwd(x, w, b, target, drow, out, rows, B, B, logits=logits)
            rpg = B
        dlog, tape = L.forward_tape(plan, zin, groups=groups, loss_fold=fold)
        self._carry.setdefault('folded', {})[which] = (fold, plan[-1], tape[-1][0], dlog.shape, rows.numel())
        return dlog, tape, rows, rpg

    def recon_logits(self):
        """(image logits, label logits) of the step that just ran.  A folded decoder never stored them: its last
        launch is re-issued -- same kernel, same accumulators -- with the optional logits output (scratch buffers for
        the rest).  The parity tests use this to find logits that are EXACTLY zero."""
        logits_lbl, _, _, logits_img, _, _ = self._carry['keep'][-1]
        out = {'image': logits_img, 'label': logits_lbl}
        for which, (fold, op, x, shape, nrows) in self._carry.get('folded', {}).items():
            w, b = L._lin_weights(op)
            lg = torch.empty(shape, dtype=torch.float32, device=self.dev)
            fold(x, w.detach(), None if b is None else b.detach(), torch.empty_like(lg), logits=lg,
                 rows=torch.empty(nrows, dtype=torch.float32, device=self.dev))
            out[which] = lg
        return out['image'], out['label']

    # ------------------------------------------------------------------ host-side setup per step
    def set_coefficients(self, annealing_factor):
        B = float(self.B)
        _, c = self.tables.begin()
        c = c.reshape(3, self.T)
        for t in range(self.T):
            c[0, t] = self.lambda_image / B if self.img_has_loss[t] else 0.0
            c[1, t] = self.lambda_label / B if self.lbl_has_loss[t] else 0.0
            c[2, t] = float(annealing_factor) / B
        self.tables.commit()

    def set_noise(self, noise):
        """Parity mode: ``noise`` = {'eps': [3 x [B,D]], 'mask': [3 x [B,512] or None]} in the
        REFERENCE's call order (joint, image, label), e.g. oracle.steps.draw_bimodal_noise."""
        self._draw_in_poe = False
        for t in range(self.T):
            self.noise[t].copy_(noise['eps'][self.ref_order[t]].to(self.dev, non_blocking=True))
        if self.has_dropout:
            self.drop_masks[0].copy_(noise['mask'][0].to(self.dev))
            self.drop_masks[1].copy_(noise['mask'][1].to(self.dev))

    def draw_noise(self):
        # launch indices counter + 0 / + 1; the counter itself advances once per step (phase A's bookkeeping).
        # eps is drawn INSIDE the PoE launch (mvae_poe_fwd_draw: same values, one launch less at the head of the
        # step; MVAE_POE_DRAW=0: a philox_fill launch); the Dropout masks are needed before it and keep theirs
        self._draw_in_poe = self.poe_draw
        if not self.poe_draw:
            K.philox_fill(self.noise, self.seed, self.counter, 0)
        if self.has_dropout:
            K.philox_fill(self.drop_masks, self.seed ^ 0x9E3779B97F4A7C15, self.counter, 1, keep_prob=KEEP)

    def terms_in_reference_order(self, elbo):
        """elbo[T+1] (engine order) -> [joint, image, label] + [total]."""
        idx = [self.ref_order.index(r) for r in range(self.T)] + [self.T]
        return elbo[idx]

    # ------------------------------------------------------------------ the step
    def _phase_a(self, image, label):
        m, B, D, T = self.model, self.B, self.D, self.T
        if image.shape[0] != B:
            raise ValueError('engine was built for batch %d, got %d' % (B, image.shape[0]))
        c = self._carry = {}
        image = image.contiguous()
        n_up = 2  # each encoder is called twice per step in the reference
        # ---- encoders: label on the side stream, image on this one
        lbl_in = label if m.LABEL_KIND == 'class' else label.float().contiguous()

        def encode_label(after=None):
            with self._branch(after):
                heads, c['tape_lbl'] = L.forward_tape(m.label_encoder.plan(), lbl_in, bn_updates=n_up)
                if self.counter_rides == 'encoder' or self.pair_dec:
                    self._early_counter()
            return heads
        paired = self._pair_enc_now()
        fork = None if paired else self._fork_point()
        if paired:
            heads_img, heads_lbl = self._encoders_paired_fwd(image, lbl_in)
            img_experts = [heads_img]
        elif fork is None:
            heads_lbl = encode_label()
        if paired:
            pass
        elif self.has_dropout:
            h, c['tape_trunk'] = L.forward_tape(self.trunk, image, groups=1, bn_updates=n_up)
            hd = torch.empty(2 * B, h.shape[1], dtype=torch.float32, device=self.dev)
            K.dropout_fanout_fwd(h, self.drop_masks, hd, 1.0 / KEEP)
            heads_img, c['tape_head'] = L.forward_tape(self.head, hd)
            img_experts = [heads_img[:B], heads_img[B:]]
        else:
            heads_img, c['tape_img'] = L.forward_tape(m.image_encoder.plan(), image, bn_updates=n_up)
            img_experts = [heads_img]
        if fork is not None:
            heads_lbl = encode_label(fork)
        self._join()
        experts = img_experts + [heads_lbl]
        mus = [e[:, :D] for e in experts]
        lvs = [e[:, D:] for e in experts]
        # ---- PoE + reparameterise + KL for all three terms
        mu = torch.empty(T, B, D, dtype=torch.float32, device=self.dev)
        lv = torch.empty_like(mu)
        z = torch.empty_like(mu)
        kl = torch.empty(T, B, dtype=torch.float32, device=self.dev)
        self._poe_forward(mus, lvs, mu, lv, z, kl)
        self.last_latents = (mu, lv, z)
        i0, ni = self.img_terms
        l0, nl = self.lbl_terms
        if self.pair_dec:
            g_img, g_lbl, rows_img, rows_lbl, keep_dec = self._decoders_paired(z, image, label, lbl_in)
            rpg_img = rpg_lbl = B
        else:
            def label_branch(after=None):
                # ---- label branch: decoder forward, reconstruction term + gradient, decoder backward
                with self._branch(after):
                    if paired or self.counter_rides == 'decoder':
                        self._early_counter()       # 'paired': no encoder side branch, the step counter rides this one
                    zl = z[l0:l0 + nl].reshape(nl * B, D)
                    lbl_kind = 'class' if m.LABEL_KIND == 'class' else 'bce'
                    rpg_lbl = B
                    if self.fold_label and self._fold_plan(m.label_decoder.plan(), lbl_kind):
                        logits_lbl = None
                        dlog_lbl, tape_dl, rows_lbl, rpg_lbl = self._decode_with_loss(
                            m.label_decoder.plan(), zl, nl, lbl_kind, label if lbl_kind == 'class' else lbl_in,
                            self.coef[1, l0:l0 + nl], 'label')
                    else:
                        logits_lbl, tape_dl = L.forward_tape(m.label_decoder.plan(), zl, groups=nl)
                        rows_lbl = torch.empty(nl * B, dtype=torch.float32, device=self.dev)
                        dlog_lbl = torch.empty_like(logits_lbl)
                        if m.LABEL_KIND == 'class':
                            K.ce_fwd(logits_lbl, label, rows_lbl, drow=self.coef[1, l0:l0 + nl], dlogits=dlog_lbl,
                                     rows_per_group=B, label_rows=B)
                        else:
                            K.bce_rowsum_fwd(logits_lbl, lbl_in, rows_lbl, drow=self.coef[1, l0:l0 + nl],
                                             dlogits=dlog_lbl, rows_per_group=B, target_rows=B)
                    wl = self._deferred()
                    if self.split_dz:
                        # this decoder's latent gradient goes to its OWN buffer (terms l0 .. l0+nl-1), on this stream
                        dz_lbl = torch.empty(nl * B, D, dtype=torch.float32, device=self.dev)
                        L.backward_tape(m.label_decoder.plan(), tape_dl, dlog_lbl, groups=nl, need_input_grad=True,
                                        input_grad_out=dz_lbl, deferred=wl)
                        g_lbl = dz_lbl
                    else:
                        g_lbl = L.backward_tape(m.label_decoder.plan(), tape_dl, dlog_lbl, groups=nl,
                                                defer_input_grad=True, deferred=wl)
                    ev_lbl = None
                    # single-GPU step, or the one-graph data-parallel step (the communicator's launch is a stream
                    # operation: bucket 0 can then go out from the side stream, behind the last decoder gradient)
                    # (with three buckets phase B's first half must then not JOIN the streams before the second: _upper_done)
                    dp_side = self._side_launch_ok() and self.n_buckets >= 2
                    if self.wgrad_on_side and self.side is not None and isinstance(wl, L.WgradBatch) \
                            and (dp_side or (self._comm is None and self.on_bucket_ready is None)):
                        ev_lbl = torch.cuda.Event()
                        ev_lbl.record()              # this decoder's latent gradient is final: the PoE backward may start
                    self._launch_deferred(wl, self.wg_side)
                return logits_lbl, tape_dl, dlog_lbl, rows_lbl, rpg_lbl, g_lbl, ev_lbl, dp_side

            def image_branch():
                # ---- image branch (this stream)
                zi = z[i0:i0 + ni].reshape(ni * B, D)
                fold_img = self.fold_image and self._fold_plan(m.image_decoder.plan(), 'bce')
                rpg_img = B
                if fold_img:
                    logits_img = None
                    dlog_img, tape_di, rows_img, rpg_img = self._decode_with_loss(
                        m.image_decoder.plan(), zi, ni, 'bce', image, self.coef[0, i0:i0 + ni], 'image')
                else:
                    logits_img, tape_di = L.forward_tape(m.image_decoder.plan(), zi, groups=ni)
                if self.has_bn and ni < T:
                    # the reference also decodes the image for the label-only call: no loss, but its
                    # BatchNorm running statistics advance (celeba/train.py:195)
                    L.forward_tape(m.image_decoder.plan(), z[i0 + ni:].reshape((T - ni) * B, D), groups=T - ni,
                                   stats_only=True)
                if not fold_img:
                    P = logits_img[0].numel()
                    li = logits_img.reshape(ni * B, P)
                    rows_img = torch.empty(ni * B, dtype=torch.float32, device=self.dev)
                    dlog_img = torch.empty_like(li)
                    K.bce_rowsum_fwd(li, image.reshape(B, P), rows_img, drow=self.coef[0, i0:i0 + ni], dlogits=dlog_img,
                                     rows_per_group=B, target_rows=B)
                    dlog_img = dlog_img.reshape(logits_img.shape)
                wi = self._deferred()
                if self.split_dz:
                    dz_img = torch.empty(ni * B, D, dtype=torch.float32, device=self.dev)
                    L.backward_tape(m.image_decoder.plan(), tape_di, dlog_img, groups=ni,
                                    need_input_grad=True, input_grad_out=dz_img, deferred=wi)
                    g_img = dz_img
                else:
                    g_img = L.backward_tape(m.image_decoder.plan(), tape_di, dlog_img,
                                            groups=ni, defer_input_grad=True, deferred=wi)
                return logits_img, tape_di, dlog_img, rows_img, rpg_img, g_img, wi

            # at the fork the main stream's continuation -- the longer chain -- is issued first (MVAE_MAIN_FIRST)
            dec_first = self.main_first_dec == '1' or (self.main_first_dec == 'auto' and any(
                op.kind in ('conv', 'convT') for op in m.image_decoder.plan()))
            fork = self._fork_point() if dec_first else None
            if fork is None:
                logits_lbl, tape_dl, dlog_lbl, rows_lbl, rpg_lbl, g_lbl, ev_lbl, dp_side = label_branch()
            logits_img, tape_di, dlog_img, rows_img, rpg_img, g_img, wi = image_branch()
            if fork is not None:
                logits_lbl, tape_dl, dlog_lbl, rows_lbl, rpg_lbl, g_lbl, ev_lbl, dp_side = label_branch(fork)
            if ev_lbl is not None and isinstance(wi, L.WgradBatch):
                # The image side is the longer chain and the label side has slack (MNIST: ~225 vs ~150 us of kernels):
                # the image decoder's weight-gradient batch -- nothing before the optimizer reads it -- goes to the
                # SIDE stream behind the label decoder's, and this stream waits only for the label decoder's latent
                # gradient (an event), not for the side stream's weight gradients.  MVAE_WGRAD_SIDE=0: both on their
                # own stream and a full join here.
                ev_img = torch.cuda.Event()
                ev_img.record()
                # the batch's gradient tensors were allocated on THIS stream and are read on the other one: they must
                # stay referenced until the streams have joined, or the allocator hands their memory to this
                # stream's next launches while the weight-gradient kernel is still queued
                c['wgrad_on_side_keep'] = list(wi)
                # (with the third stream of the single-GPU step this batch was tried there as well: MNIST +10 %, CelebA +0.7 %,
                #  profiles/r06_sched_ab.txt -- it stays on the side stream, behind the label decoder's chain)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ev_img)
                    wi.flush()
                    if getattr(self, '_adam_split', None) is not None:
                        ev_dec = torch.cuda.Event()
                        ev_dec.record()              # every decoder weight gradient is final behind this
                        c['ev_dec_grads'] = ev_dec
                    if dp_side:
                        self._comm.launch(0)         # every decoder gradient is final here, on THIS stream
                        self._bucket0_done = True
                torch.cuda.current_stream(self.dev).wait_event(ev_lbl)
                c['events'] = (ev_lbl, ev_img)
            else:
                self._launch_deferred(wi, self.wg_main)     # decoder weight gradients run behind phase B
                self._join()
            keep_dec = (logits_lbl, tape_dl, dlog_lbl, logits_img, tape_di, dlog_img)
        # ---- ELBO per term and total (mnist/train.py:57-58,214), the cleared dz and the step's Philox counter
        #      advance: one bookkeeping launch
        elbo_parts = [(kl, self.coef[2], None, 0, T, B),
                      (rows_img, self.coef[0, i0:i0 + ni], None, i0, ni, rpg_img),
                      (rows_lbl, self.coef[1, l0:l0 + nl], None, l0, nl, rpg_lbl)]
        counter_inc = 2 if self.has_dropout else 1
        if self.split_dz and not self.pair_dec:
            # Each decoder's backward ran to its input on its own stream into its own latent-gradient buffer;
            # poe_bwd adds the two per term (image first: the bits of accumulating into one cleared buffer).
            # Nothing is left between the join and the PoE backward: the ELBO bookkeeping launch moves behind the
            # image encoder's backward, where the main stream waits for the label side anyway.
            dz = None
            c['dz_split'] = (g_img, [t - i0 if i0 <= t < i0 + ni else -1 for t in range(T)],
                             g_lbl, [t - l0 if l0 <= t < l0 + nl else -1 for t in range(T)])
            c['elbo_late'] = (elbo_parts, counter_inc)
        else:
            dz = torch.empty(T, B, D, dtype=torch.float32, device=self.dev)
            K.elbo_reduce(elbo_parts, self.elbo, T, zero=dz, counter_dev=self.counter, counter_inc=counter_inc)
            # ---- both decoders' first layers -> the shared dz
            L.first_linear_dgrad(m.image_decoder.plan(), g_img, dz[i0:i0 + ni].reshape(ni * B, D), True)
            L.first_linear_dgrad(m.label_decoder.plan(), g_lbl, dz[l0:l0 + nl].reshape(nl * B, D), True)
        # everything a branch allocated stays referenced until the next step's first fork
        c.update(mus=mus, lvs=lvs, mu=mu, lv=lv, dz=dz, heads_img=heads_img, heads_lbl=heads_lbl,
                 keep=(z, kl, rows_lbl, g_lbl, rows_img, g_img, lbl_in, keep_dec))
        if self._comm is not None or self.on_bucket_ready is not None:
            self._join_wgrad()      # data parallel: the decoder bucket is all-reduced after phase A

    def _late_elbo(self):
        """The step's ELBO sums + Philox counter advance (split-dz mode): off the decoder -> PoE -> encoder chain."""
        late = self._carry.pop('elbo_late', None)
        if late is not None:
            K.elbo_reduce(late[0], self.elbo, self.T, counter_dev=self.counter, counter_inc=late[1])

    def _phase_b(self, part='all'):
        """PoE backward + encoders backward.  ``part``: 'all', or for a data-parallel replica with three
        gradient buckets 'upper' (everything but the image encoder's first layers -- the arena tail) then
        'lower' (those layers), so the second bucket's all-reduce starts before the last convs are done."""
        m, B, D = self.model, self.B, self.D
        c = self._carry
        if part in ('all', 'upper'):
            heads_img, heads_lbl = c['heads_img'], c['heads_lbl']
            # ---- PoE backward -> encoder heads
            g_heads_img = torch.empty_like(heads_img)
            g_heads_lbl = torch.empty_like(heads_lbl)
            g_list = ([g_heads_img[:B], g_heads_img[B:]] if self.has_dropout else [g_heads_img]) + [g_heads_lbl]
            if c.get('dz_split') is not None:
                dza, sa, dzb, sb = c['dz_split']
                K.poe_bwd_split(c['mus'], c['lvs'], self.masks_dev, self.noise, c['mu'], c['lv'], dza, sa, dzb, sb,
                                self.coef[2], [gg[:, :D] for gg in g_list], [gg[:, D:] for gg in g_list],
                                m.POE_VARIANT, dkl_per_term=True)
            else:
                K.poe_bwd(c['mus'], c['lvs'], self.masks_dev, self.noise, c['mu'], c['lv'], c['dz'], None, None,
                          self.coef[2], [gg[:, :D] for gg in g_list], [gg[:, D:] for gg in g_list], m.POE_VARIANT,
                          dkl_per_term=True)
            c['keep_b'] = (g_heads_img, g_heads_lbl)
        if part == 'all' and 'enc_pair' in c:
            self._encoders_paired_bwd(*c['keep_b'])
            self._late_elbo()
            self._join()
            self._join_wgrad()
            return
        if part == 'all':
            # ---- encoders backward: data-gradient chains on the two branches, then ALL their weight
            #      gradients spread over this stream and the two weight-gradient streams
            wl, wi = self._deferred(), self._deferred()

            def label_encoder_backward(after=None):
                with self._branch(after):
                    L.backward_tape(m.label_encoder.plan(), c['tape_lbl'], g_heads_lbl, deferred=wl)
                    if isinstance(wl, L.WgradBatch):
                        wl.flush()
            fork = self._fork_point()
            if fork is None:
                label_encoder_backward()
            if self.has_dropout:
                d_hd = L.backward_tape(self.head, c['tape_head'], g_heads_img, need_input_grad=True, deferred=wi)
                d_h = torch.empty(B, d_hd.shape[1], dtype=torch.float32, device=self.dev)
                K.dropout_fanin_bwd(d_hd, self.drop_masks, d_h, 1.0 / KEEP)
                L.backward_tape(self.trunk, c['tape_trunk'], d_h, deferred=wi)
            else:
                L.backward_tape(m.image_encoder.plan(), c['tape_img'], g_heads_img, deferred=wi)
            if isinstance(wi, L.WgradBatch):
                if (os.environ.get('MVAE_TAIL_LANES', '1') == '1' and self.wg_batched and self.wg_main is not None
                        and self._comm is None and self.on_bucket_ready is None):
                    # the image encoder's batch is the step's tail -- on CelebA 140 us of conv weight gradients and their finish
                    # launches one after the other, nothing beside them: every other conv closure goes to the third stream
                    # (CelebA -0.25 %, 3 of 3 interleaved rounds, profiles/r06_sched_ab.txt; MNIST has no closures)
                    lane, keep, k = L.WgradBatch(adam=wi.adam), [], 0
                    for e in list(wi):
                        if not isinstance(e, tuple):
                            (lane if k % 2 else keep).append(e)
                            k += 1
                        else:
                            keep.append(e)
                    wi[:] = keep
                    self._launch_deferred(lane, self.wg_main)
                wi.flush()
                wi = None
            if fork is not None:
                label_encoder_backward(fork)
            self._late_elbo()
            if getattr(self, '_adam_split', None) is not None and c.get('ev_dec_grads') is not None:
                opt, split, _ = self._adam_split
                torch.cuda.current_stream(self.dev).wait_event(c['ev_dec_grads'])
                opt.step_counted_range(0, split)     # the main stream waits for the side stream here anyway
                c['adam_dec_done'] = True
            self._join()
            if wi is not None:
                fns = wi + wl
                n = len(fns)
                self._launch_deferred(fns[:n // 3], self.wg_main)
                self._launch_deferred(fns[n // 3:2 * n // 3], self.wg_side)
                for fn in fns[2 * n // 3:]:
                    fn()
                c.setdefault('deferred', []).append(fns)
            self._join_wgrad()
            return
        # ---- three buckets: the image encoder's stack is cut where the arena tail begins
        plan, tape = (self.trunk, c['tape_trunk']) if self.has_dropout else (m.image_encoder.plan(), c['tape_img'])
        cut = L.tail_cut(plan, m.arena_tail())
        if part == 'upper':
            with self._branch():
                L.backward_tape(m.label_encoder.plan(), c['tape_lbl'], c['keep_b'][1])
            g = c['keep_b'][0]
            if self.has_dropout:
                d_hd = L.backward_tape(self.head, c['tape_head'], g, need_input_grad=True)
                g = torch.empty(B, d_hd.shape[1], dtype=torch.float32, device=self.dev)
                K.dropout_fanin_bwd(d_hd, self.drop_masks, g, 1.0 / KEEP)
                c['keep_b'] += (d_hd,)
            c['g_cut'] = L.backward_tape(plan[cut:], tape[cut:], g, need_input_grad=True)
            self._late_elbo()
            self._upper_done()
        else:
            L.backward_tape(plan[:cut], tape[:cut], c['g_cut'])


# =====================================================================================
# CelebA-19
# =====================================================================================
N_ATTRS = 18


def unrank_combination(n, k, index):
    """Members of the ``index``-th k-subset of range(n) in the order ``itertools.combinations`` yields
    them (lexicographic) -- the row order of the reference's pool within one subset size
    (celeba19/train.py:99-101)."""
    from math import comb
    if not 0 <= index < comb(n, k):
        raise IndexError('combination index %d out of range for C(%d,%d)' % (index, n, k))
    members, first = [], 0
    for slot in range(k, 0, -1):
        # the block of subsets whose next member is ``first`` has C(n-first-1, slot-1) rows
        while True:
            block = comb(n - first - 1, slot - 1)
            if index < block:
                break
            index -= block
            first += 1
        members.append(first)
        first += 1
    return members


def sample_subsets(rng, n_modalities=19, size=1):
    """``size`` modality subsets, draw for draw what ``sample_combinations(enumerate_combinations(n),
    size)`` of celeba19/train.py:111-142 returns under the same generator state -- without the
    524,267 x 19 pool (:87-108).  The reference makes exactly these generator calls:
      1. ``choice(pool_space, size, replace=True)`` -- a subset SIZE per sample, pool_space = 2..n-1;
      2. for each size k that was drawn c > 0 times, in increasing k:
         ``choice(range(C(n,k)), size=c, replace=False)`` -- c distinct row numbers of the size-k block,
    and a row number is the lexicographic rank of the subset (``unrank_combination``).  ``rng`` is
    ``numpy.random`` (the reference's global generator) or a ``RandomState``.  Rows come back grouped by
    size, in draw order within a size, like the reference's ``np.concatenate``."""
    import numpy as np
    from math import comb
    n = int(n_modalities)
    if size <= 0:
        return np.zeros((0, n), dtype=bool)
    sizes = rng.choice(np.arange(2, n), size, replace=True)
    counts = np.bincount(sizes, minlength=n)
    out = []
    for k in range(n):
        if counts[k] > 0:
            for r in rng.choice(comb(n, k), size=int(counts[k]), replace=False):
                row = np.zeros(n, dtype=bool)
                row[unrank_combination(n, k, int(r))] = True
                out.append(row)
    return np.stack(out)


class Celeba19Step(_StepBase):
    """Fused step for the 19-modality MVAE: complete + image-only + 18 single-attribute + M
    sampled-subset ELBO terms (celeba19/train.py:257-308) as one batched pass.

    Static launch structure, dynamic content: which attributes / whether the image enter a
    sampled term only changes device tables (PoE masks, loss coefficients, the BatchNorm
    update count), so the captured graph is valid for every step.

    experts : image draw k (k = 0: complete term, 1: image-only term, 2+j: sampled term j --
              they differ only in the Dropout mask), then the 18 attribute encoders (each runs
              once; the reference re-runs encoder i in every call that contains attribute i).
    decoders: the image decoder runs for every term in term order (its BatchNorm running
              statistics advance 20+M times per step in the reference) but
              keeps activations only for the terms whose image BCE enters an ELBO;
              attribute decoder i runs once on the gathered rows of (complete, sampled..., single i).
    """

    def __init__(self, model, batch_size, lambda_image=1.0, lambda_attrs=1.0, approx_m=1, seed=0,
                 combo_seed=681307, faithful_bn_stats=True, rng=None):
        import numpy as np
        self._init_common(model, batch_size, seed)
        self.lambda_image, self.lambda_attrs = float(lambda_image), float(lambda_attrs)
        self.M = int(approx_m)
        self.T = 2 + N_ATTRS + self.M
        self.n_img = 2 + self.M
        self.S = self.M + 2                       # decoder slots: complete, sampled..., single
        if self.n_img + N_ATTRS >= 32 or self.T > 40:      # expert bits live in a signed 32-bit mask word
            raise ValueError('approx_m too large for the PoE kernel limits (experts <= 31, terms <= 40): approx_m <= 11')
        self.faithful = bool(faithful_bn_stats)
        # subsets come from ``rng`` (``numpy.random`` = the reference's global generator: the same
        # ``np.random.seed`` then gives the reference's subsets draw for draw) or a private
        # RandomState(combo_seed) -- the same seed on every rank: same subsets, equal work
        self.rng = rng if rng is not None else np.random.RandomState(combo_seed)
        B, D, dev, T, S = self.B, self.D, self.dev, self.T, self.S
        enc = model.image_encoder
        self.trunk = L.compile_plan(enc.trunk_modules())
        self.head = L.compile_plan(enc.head_modules())
        self.enc_plans = [e.plan() for e in model.attr_encoders]
        self.dec_plans = [d.plan() for d in model.attr_decoders]
        # the 18 experts of each kind as ONE grouped launch per layer (MVAE_GROUPED=0: 18 launches)
        self.grouped = os.environ.get('MVAE_GROUPED', '1') != '0'
        if self.grouped:
            self.enc_group = L.GroupedPlans(self.enc_plans)
            self.dec_group = L.GroupedPlans(self.dec_plans)
        # device tables: [masks int32 T | n_img int32 1 | coef f32 3*T (rows: image, unused, kl) | coef_attr f32 18*S]
        self.tables = tb = StepTables(T + 1 + 3 * T + N_ATTRS * S, dev)
        self._off = (0, T, T + 1, T + 1 + 3 * T)
        self.masks_dev = tb.ints(0, T)
        self.nimg_dev = tb.ints(T, 1)
        self.coef = tb.floats(T + 1, 3 * T).reshape(3, T)
        self.coef_attr = tb.floats(T + 1 + 3 * T, N_ATTRS * S)
        self._beta = 1.0
        term_of = torch.zeros(N_ATTRS, S, dtype=torch.int32)
        for i in range(N_ATTRS):
            term_of[i, 0] = 0
            for j in range(self.M):
                term_of[i, 1 + j] = 2 + N_ATTRS + j
            term_of[i, S - 1] = 2 + i
        self.term_of_slot = term_of.reshape(-1).to(dev)
        self.noise = torch.empty(T, B, D, dtype=torch.float32, device=dev)
        self.drop_masks = torch.ones(self.n_img, B, 512, dtype=torch.float32, device=dev)
        self.elbo = torch.zeros(T + 1, dtype=torch.float32, device=dev)
        self.combos = None
        # tables start from a FIXED placeholder ({image, attribute j} for sampled term j): the constructor must not
        # draw from ``rng`` -- with ``rng = numpy.random`` every engine built (also the ragged-last-batch ones built
        # mid-epoch) would otherwise consume a draw the reference never makes, and the subsets would no longer be
        # the reference's draw for draw.  step() / replay() draw the step's real subsets.
        first = np.zeros((self.M, 1 + N_ATTRS), dtype=bool)
        first[:, 0] = True
        first[np.arange(self.M), 1 + np.arange(self.M) % N_ATTRS] = True
        self.set_terms(first)

    # ------------------------------------------------------------------ host-side tables
    def set_terms(self, combos, commit=True):
        """``combos``: bool [M, 19] (column 0 = image) -- the step's sampled subsets.  ``commit=False``:
        only remember them; the tables go to the device with the next ``set_coefficients``."""
        import numpy as np
        self.combos = np.asarray(combos, dtype=bool).reshape(self.M, 1 + N_ATTRS)
        self.n_img_present = 2 + int(self.combos[:, 0].sum())
        if commit:
            self.set_coefficients(self._beta)

    def set_coefficients(self, annealing_factor):
        """Fill one pinned slot with every per-step table and queue its copy (one DMA per step)."""
        import numpy as np
        B, T, S, M, n_img = float(self.B), self.T, self.S, self.M, self.n_img
        combos = self.combos
        self._beta = float(annealing_factor)
        wi, wf = self.tables.begin()
        o_mask, o_nimg, o_coef, o_attr = self._off
        attr_bits = (combos[:, 1:].astype(np.int64) << (n_img + np.arange(N_ATTRS))).sum(axis=1)
        m = np.zeros(T, dtype=np.int64)
        m[0] = 1 | (((1 << N_ATTRS) - 1) << n_img)            # complete: image draw 0 + all attributes
        m[1] = 1 << 1                                         # image only: image draw 1
        m[2:2 + N_ATTRS] = 1 << (n_img + np.arange(N_ATTRS))
        m[2 + N_ATTRS:] = attr_bits + np.where(combos[:, 0], 1 << (2 + np.arange(M)), 0)
        wi[o_mask:o_mask + T] = m.astype(np.int32)
        wi[o_nimg] = self.n_img_present
        c = wf[o_coef:o_coef + 3 * T].reshape(3, T)
        c[:] = 0.0
        c[0, 0] = c[0, 1] = self.lambda_image / B
        c[0, 2 + N_ATTRS:] = np.where(combos[:, 0], 1.0 / B, 0.0)   # sampled terms omit the lambdas -> 1.0 (:294-300)
        c[2, :] = self._beta / B
        ca = wf[o_attr:o_attr + N_ATTRS * S].reshape(N_ATTRS, S)
        ca[:, 0] = self.lambda_attrs / B                      # complete term uses lambda_attrs (:265-267)
        ca[:, 1:1 + M] = np.where(combos[:, 1:].T, 1.0 / B, 0.0)
        ca[:, S - 1] = 1.0 / B                                # single-attribute terms omit it (:281-282)
        self.tables.commit()

    def set_noise(self, noise):
        """``noise`` in the reference's term order (oracle.steps.draw_celeba19_noise)."""
        self._draw_in_poe = False
        for t in range(self.T):
            self.noise[t].copy_(noise['eps'][t].to(self.dev, non_blocking=True))
        img_terms = [0, 1] + [2 + N_ATTRS + j for j in range(self.M)]
        for k, t in enumerate(img_terms):
            mk = noise['mask'][t]
            if mk is not None:
                self.drop_masks[k].copy_(mk.to(self.dev))
            else:
                self.drop_masks[k].fill_(1.0)      # expert masked out of the PoE: value irrelevant

    def draw_noise(self):
        self._draw_in_poe = self.poe_draw
        if not self.poe_draw:
            K.philox_fill(self.noise, self.seed, self.counter, 0)
        K.philox_fill(self.drop_masks, self.seed ^ 0x9E3779B97F4A7C15, self.counter, 1, keep_prob=KEEP)

    def step(self, image, attrs, annealing_factor, noise=None, combos=None):
        self.set_terms(combos if combos is not None else sample_subsets(self.rng, 1 + N_ATTRS, self.M), commit=False)
        return _StepBase.step(self, image, attrs, annealing_factor, noise=noise)

    def replay(self, image, attrs, annealing_factor, combos=None):
        self.set_terms(combos if combos is not None else sample_subsets(self.rng, 1 + N_ATTRS, self.M), commit=False)
        return _StepBase.replay(self, image, attrs, annealing_factor)

    def terms_in_reference_order(self, elbo):
        return elbo

    def _bump_bn_counters(self):
        # the image-encoder trunk's BatchNorms advance once per term that contains the image,
        # which varies with the sampled subsets; everything else is static
        trunk = set(id(op.mod) for op in self.trunk if op.kind == 'bn')
        for m, inc in self._bn_inc:
            m._nbt_pending += self.n_img_present if id(m) in trunk else inc

    # ------------------------------------------------------------------ the step
    def _phase_a(self, image, attrs):
        m, B, D, T, S, M, n_img = self.model, self.B, self.D, self.T, self.S, self.M, self.n_img
        dev = self.dev
        c = self._carry = {}
        image = image.contiguous()
        attrs = attrs.float().contiguous()
        # ---- 18 attribute encoders, each once (no BatchNorm / Dropout: celeba19/model.py:173-178),
        #      on the side stream: ~110 small launches that hide behind the image encoder
        heads_attr, c['tape_enc'] = [], []
        with self._branch():
            if self.grouped:
                heads_all, c['tape_enc'] = L.forward_tape_grouped(self.enc_group, attrs)
                heads_attr = [heads_all[i] for i in range(N_ATTRS)]
            else:
                for i in range(N_ATTRS):
                    ha, tp = L.forward_tape(self.enc_plans[i], attrs[:, i])
                    heads_attr.append(ha); c['tape_enc'].append(tp)
            self._early_counter()
        # ---- image encoder: trunk once, n_img Dropout draws, head on n_img*B rows
        h, c['tape_trunk'] = L.forward_tape(self.trunk, image, bn_updates=self.n_img_present,
                                            bn_updates_dev=self.nimg_dev)
        hd = torch.empty(n_img * B, h.shape[1], dtype=torch.float32, device=dev)
        K.dropout_fanout_fwd(h, self.drop_masks, hd, 1.0 / KEEP)
        heads_img, c['tape_head'] = L.forward_tape(self.head, hd)
        self._join()
        experts = [heads_img[k * B:(k + 1) * B] for k in range(n_img)] + heads_attr
        mus = [e[:, :D] for e in experts]
        lvs = [e[:, D:] for e in experts]
        mu = torch.empty(T, B, D, dtype=torch.float32, device=dev)
        lv = torch.empty_like(mu); z = torch.empty_like(mu)
        kl = torch.empty(T, B, dtype=torch.float32, device=dev)
        self._poe_forward(mus, lvs, mu, lv, z, kl)
        self.last_latents = (mu, lv, z)
        t_s = 2 + N_ATTRS
        # ---- attribute decoders (side stream): gather the z rows each one needs, one pass per
        #      decoder, reconstruction term + gradient, backward into dzcat
        with self._branch():
            zcat = torch.empty(N_ATTRS, S * B, D, dtype=torch.float32, device=dev)
            K.block_gather(z, self.term_of_slot, zcat, B * D)
            logits_attr = torch.empty(N_ATTRS * S, B, dtype=torch.float32, device=dev)
            tape_dec = []
            if self.grouped:
                _, tape_dec = L.forward_tape_grouped(self.dec_group, zcat, final_out=logits_attr)
            else:
                for i in range(N_ATTRS):
                    _, tp = L.forward_tape(self.dec_plans[i], zcat[i], final_out=logits_attr[i * S:(i + 1) * S])
                    tape_dec.append(tp)
            rows_attr = torch.empty(N_ATTRS * S, dtype=torch.float32, device=dev)
            dlog_attr = torch.empty_like(logits_attr)
            # logits row (i, s) holds B columns; its targets are column i of attrs[B, 18]
            K.bce_rowsum_fwd(logits_attr, attrs, rows_attr, drow=self.coef_attr, dlogits=dlog_attr,
                             rows_per_group=1, target_rows=N_ATTRS, target_div=S, target_strides=(1, N_ATTRS))
            dzcat = torch.empty_like(zcat)
            if self.grouped:
                L.backward_tape_grouped(self.dec_group, tape_dec, dlog_attr.reshape(N_ATTRS, S * B, 1),
                                        need_input_grad=True, input_grad_out=dzcat)
            else:
                for i in range(N_ATTRS):
                    L.backward_tape(self.dec_plans[i], tape_dec[i], dlog_attr[i * S:(i + 1) * S].reshape(S * B, 1),
                                    need_input_grad=True, input_grad_out=dzcat[i], input_grad_accumulate=False)
        # ---- image decoder in term order: [0,1] kept, [2..19] statistics only, sampled kept
        dplan = m.image_decoder.plan()
        logit_a, tape_a = L.forward_tape(dplan, z[0:2].reshape(2 * B, D), groups=2)
        if self.faithful:
            L.forward_tape(dplan, z[2:t_s].reshape(N_ATTRS * B, D), groups=N_ATTRS, stats_only=True)
        if M > 0:
            logit_c, tape_c = L.forward_tape(dplan, z[t_s:T].reshape(M * B, D), groups=M)
        P = logit_a[0].numel()
        img_flat = image.reshape(B, P)
        rows_a = torch.empty(2 * B, dtype=torch.float32, device=dev)
        dlog_a = torch.empty(2 * B, P, dtype=torch.float32, device=dev)
        K.bce_rowsum_fwd(logit_a.reshape(2 * B, P), img_flat, rows_a, drow=self.coef[0, 0:2], dlogits=dlog_a,
                         rows_per_group=B, target_rows=B)
        rows_c = dlog_c = None
        if M > 0:
            rows_c = torch.empty(M * B, dtype=torch.float32, device=dev)
            dlog_c = torch.empty(M * B, P, dtype=torch.float32, device=dev)
            K.bce_rowsum_fwd(logit_c.reshape(M * B, P), img_flat, rows_c, drow=self.coef[0, t_s:T],
                             dlogits=dlog_c, rows_per_group=B, target_rows=B)
        # ---- image decoder backward -> dz (only this stream touches dz before the join)
        dz = torch.empty(T, B, D, dtype=torch.float32, device=dev)
        K.fill_(dz, 0.0)
        L.backward_tape(dplan, tape_a, dlog_a.reshape(logit_a.shape), need_input_grad=True, groups=2,
                        input_grad_out=dz[0:2].reshape(2 * B, D), input_grad_accumulate=True)
        if M > 0:
            L.backward_tape(dplan, tape_c, dlog_c.reshape(logit_c.shape), need_input_grad=True, groups=M,
                            input_grad_out=dz[t_s:T].reshape(M * B, D), input_grad_accumulate=True)
        self._join()
        K.block_scatter_add(dzcat, self.term_of_slot, dz, T, B * D)
        # ---- ELBO per term and total (celeba19/train.py:59,265-302) + the Philox counter advance: one launch
        parts = [(kl, self.coef[2], None, 0, T, B), (rows_a, self.coef[0, 0:2], None, 0, 2, B)]
        if M > 0:
            parts.append((rows_c, self.coef[0, t_s:T], None, t_s, M, B))
        parts.append((rows_attr, self.coef_attr, self.term_of_slot, 0, N_ATTRS * S, 1))
        K.elbo_reduce(parts, self.elbo, T, counter_dev=self.counter, counter_inc=2)
        c.update(mus=mus, lvs=lvs, mu=mu, lv=lv, dz=dz, heads_img=heads_img, heads_attr=heads_attr,
                 keep=(z, kl, zcat, logits_attr, tape_dec, rows_attr, dlog_attr, dzcat, attrs, image,
                       rows_a, rows_c, dlog_a, dlog_c))

    def _phase_b(self, part='all'):
        m, B, D, n_img = self.model, self.B, self.D, self.n_img
        c = self._carry
        if part in ('all', 'upper'):
            g_img = torch.empty_like(c['heads_img'])
            g_attr_all = torch.empty(N_ATTRS, B, 2 * D, dtype=torch.float32, device=self.dev)
            g_attr = [g_attr_all[i] for i in range(N_ATTRS)]
            g_list = [g_img[k * B:(k + 1) * B] for k in range(n_img)] + g_attr
            K.poe_bwd(c['mus'], c['lvs'], self.masks_dev, self.noise, c['mu'], c['lv'], c['dz'], None, None,
                      self.coef[2], [g[:, :D] for g in g_list], [g[:, D:] for g in g_list], m.POE_VARIANT,
                      dkl_per_term=True)
            with self._branch():
                if self.grouped:
                    L.backward_tape_grouped(self.enc_group, c['tape_enc'], g_attr_all)
                else:
                    for i in range(N_ATTRS):
                        L.backward_tape(self.enc_plans[i], c['tape_enc'][i], g_attr[i])
            d_hd = L.backward_tape(self.head, c['tape_head'], g_img, need_input_grad=True)
            d_h = torch.empty(B, d_hd.shape[1], dtype=torch.float32, device=self.dev)
            K.dropout_fanin_bwd(d_hd, self.drop_masks, d_h, 1.0 / KEEP)
            c['keep_b'] = (g_img, g_attr_all, d_hd, d_h)
            if part == 'all':
                L.backward_tape(self.trunk, c['tape_trunk'], d_h)
            else:   # data parallel, three buckets: stop where the arena tail (the conv stack) begins
                cut = L.tail_cut(self.trunk, m.arena_tail())
                c['g_cut'] = L.backward_tape(self.trunk[cut:], c['tape_trunk'][cut:], d_h, need_input_grad=True)
            if part == 'upper':
                self._upper_done()
            else:
                self._join()
        else:
            cut = L.tail_cut(self.trunk, m.arena_tail())
            L.backward_tape(self.trunk[:cut], c['tape_trunk'][:cut], c['g_cut'])

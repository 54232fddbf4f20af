"""Fused MVAE train step: the body of the reference's ``train(epoch)`` closure
(mnist/train.py:197-218, celeba/train.py:190-212) as ONE batched pass instead of three
``model()`` calls.

What the reference does per step: 2x image encoder, 2x label encoder,
3x image decoder, 3x label decoder, 3x PoE + reparameterise, 3 ELBOs, one backward.  Here:

  * each encoder runs ONCE -- its output is identical in every call that includes the modality
    (BatchNorm sees the same batch, so the same statistics; the running statistics are advanced
    twice, as in the reference).  CelebA's image encoder differs between its two calls only in
    the Dropout(0.1) draw: the conv trunk + Linear(6400,512) + Swish run once, the two masks
    are applied by a fan-out kernel and the final Linear runs on 2B rows;
  * the three PoE / reparameterise / KL evaluations are one launch over T = 3 terms;
  * each decoder runs once on the rows of all the terms that need it, BatchNorm statistics
    per term (``groups``), running statistics advanced in the reference's call order; decoder
    outputs the reference computes but never uses (mnist/train.py:208,211) are only evaluated
    where they have a side effect (BatchNorm running statistics -- SURVEY.md Appendix B-4);
  * the BCE / CE kernels emit the loss rows and d loss / d logits in one pass
    (d ELBO / d row is the constant lambda / B);
  * the backward is explicit (``layers.backward_tape``), weight gradients land in the
    gradient arena, and the whole step is a fixed launch sequence that ``capture()`` records
    into a hipGraph (``torch.cuda.CUDAGraph``) together with the optimizer.

Results (per-term ELBOs, total, every gradient, BatchNorm running statistics) equal the
reference's three-call step on the same noise; ``tests/test_engine_gpu.py`` checks that against
the oracle and the golden fixtures.
"""
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


class BimodalStep(object):
    """Fused step for ``mnist`` / ``fashionmnist`` / ``celeba`` MVAEs."""

    def __init__(self, model, batch_size, lambda_image=1.0, lambda_label=1.0, seed=0):
        model.finalize()
        self.model = model
        self.B = int(batch_size)
        self.D = model.n_latents
        self.lambda_image = float(lambda_image)
        self.lambda_label = float(lambda_label)
        self.dev = next(model.parameters()).device
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
        # per-term loss coefficients lambda/B, beta/B: pinned host mirror -> device, so a captured
        # graph sees new annealing factors without re-capture
        self.coef_host = torch.zeros(3, self.T, dtype=torch.float32).pin_memory() \
            if torch.cuda.is_available() else torch.zeros(3, self.T)
        self.coef = torch.zeros(3, self.T, dtype=torch.float32, device=dev)
        self.noise = torch.empty(self.T, B, self.D, dtype=torch.float32, device=dev)
        self.drop_masks = torch.empty(2, B, 512, dtype=torch.float32, device=dev) if self.has_dropout else None
        self.elbo = torch.zeros(self.T + 1, dtype=torch.float32, device=dev)
        self.seed = int(seed) or 1
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.on_bucket_ready = None     # parallel.py hooks gradient all-reduce launches here
        self._graph = None

    # ------------------------------------------------------------------ host-side setup per step
    def set_coefficients(self, annealing_factor):
        B = float(self.B)
        for t in range(self.T):
            self.coef_host[0, t] = self.lambda_image / B if self.img_has_loss[t] else 0.0
            self.coef_host[1, t] = self.lambda_label / B if self.lbl_has_loss[t] else 0.0
            self.coef_host[2, t] = float(annealing_factor) / B
        self.coef.copy_(self.coef_host, non_blocking=True)

    def set_noise(self, noise):
        """Parity mode: ``noise`` = {'eps': [3 x [B,D]], 'mask': [3 x [B,512] or None]} in the
        REFERENCE's call order (joint, image, label), e.g. oracle.steps.draw_bimodal_noise."""
        for t in range(self.T):
            self.noise[t].copy_(noise['eps'][self.ref_order[t]].to(self.dev, non_blocking=True))
        if self.has_dropout:
            self.drop_masks[0].copy_(noise['mask'][0].to(self.dev))
            self.drop_masks[1].copy_(noise['mask'][1].to(self.dev))

    def draw_noise(self):
        K.randn_(self.noise, self.seed, self.counter)
        if self.has_dropout:
            K.bernoulli_(self.drop_masks, KEEP, self.seed ^ 0x9E3779B97F4A7C15, self.counter)

    # ------------------------------------------------------------------ the step
    def forward_backward(self, image, label):
        """Launch the whole forward + backward.  Gradients go to ``p.grad`` (the arena); returns
        the device tensor ``elbo[T+1]`` = per-term ELBOs (engine order) and their sum."""
        m, B, D, T = self.model, self.B, self.D, self.T
        if image.shape[0] != B:
            raise ValueError('engine was built for batch %d, got %d' % (B, image.shape[0]))
        image = image.contiguous()
        n_up = 2  # each encoder is called twice per step in the reference
        # ---- encoders
        if self.has_dropout:
            h, tape_trunk = L.forward_tape(self.trunk, image, groups=1, bn_updates=n_up)
            hd = torch.empty(2 * B, h.shape[1], dtype=torch.float32, device=self.dev)
            K.dropout_fanout_fwd(h, self.drop_masks, hd, 1.0 / KEEP)
            heads_img, tape_head = L.forward_tape(self.head, hd)
            img_experts = [heads_img[:B], heads_img[B:]]
        else:
            heads_img, tape_img = L.forward_tape(m.image_encoder.plan(), image, bn_updates=n_up)
            img_experts = [heads_img]
        lbl_in = label if m.LABEL_KIND == 'class' else label.float().contiguous()
        heads_lbl, tape_lbl = L.forward_tape(m.label_encoder.plan(), lbl_in, bn_updates=n_up)
        experts = img_experts + [heads_lbl]
        mus = [e[:, :D] for e in experts]
        lvs = [e[:, D:] for e in experts]
        # ---- PoE + reparameterise + KL for all three terms
        mu = torch.empty(T, B, D, dtype=torch.float32, device=self.dev)
        lv = torch.empty_like(mu)
        z = torch.empty_like(mu)
        kl = torch.empty(T, B, dtype=torch.float32, device=self.dev)
        K.poe_fwd(mus, lvs, self.masks_dev, self.noise, mu, lv, z, kl, m.POE_VARIANT)
        self.last_latents = (mu, lv, z)
        # ---- decoders
        i0, ni = self.img_terms
        l0, nl = self.lbl_terms
        zi = z[i0:i0 + ni].reshape(ni * B, D)
        logits_img, tape_di = L.forward_tape(m.image_decoder.plan(), zi, groups=ni)
        if self.has_bn and ni < T:
            # the reference also decodes the image for the label-only call: no loss, but its
            # BatchNorm running statistics advance (celeba/train.py:195)
            L.forward_tape(m.image_decoder.plan(), z[i0 + ni:].reshape((T - ni) * B, D), groups=T - ni)
        zl = z[l0:l0 + nl].reshape(nl * B, D)
        logits_lbl, tape_dl = L.forward_tape(m.label_decoder.plan(), zl, groups=nl)
        # ---- reconstruction terms, forward and gradient in one pass each
        P = logits_img[0].numel()
        li = logits_img.reshape(ni * B, P)
        rows_img = torch.empty(ni * B, dtype=torch.float32, device=self.dev)
        dlog_img = torch.empty_like(li)
        K.bce_rowsum_fwd(li, image.reshape(B, P), rows_img, drow=self.coef[0, i0:i0 + ni], dlogits=dlog_img,
                         rows_per_group=B, target_rows=B)
        rows_lbl = torch.empty(nl * B, dtype=torch.float32, device=self.dev)
        dlog_lbl = torch.empty_like(logits_lbl)
        if m.LABEL_KIND == 'class':
            K.ce_fwd(logits_lbl, label, rows_lbl, drow=self.coef[1, l0:l0 + nl], dlogits=dlog_lbl,
                     rows_per_group=B, label_rows=B)
        else:
            K.bce_rowsum_fwd(logits_lbl, lbl_in, rows_lbl, drow=self.coef[1, l0:l0 + nl], dlogits=dlog_lbl,
                             rows_per_group=B, target_rows=B)
        # ---- ELBO per term and total (mnist/train.py:57-58,214)
        elbo = self.elbo
        K.group_sums(kl, self.coef[2], elbo[:T], elbo[T:], T, B, accumulate=False)
        K.group_sums(rows_img, self.coef[0, i0:i0 + ni], elbo[i0:i0 + ni], elbo[T:], ni, B, accumulate=True)
        K.group_sums(rows_lbl, self.coef[1, l0:l0 + nl], elbo[l0:l0 + nl], elbo[T:], nl, B, accumulate=True)
        # ---- backward: decoders -> dz
        dz = torch.empty(T, B, D, dtype=torch.float32, device=self.dev)
        K.fill_(dz, 0.0)
        g = L.backward_tape(m.image_decoder.plan(), tape_di, dlog_img.reshape(logits_img.shape),
                            need_input_grad=True, groups=ni,
                            input_grad_out=dz[i0:i0 + ni].reshape(ni * B, D), input_grad_accumulate=True)
        g = L.backward_tape(m.label_decoder.plan(), tape_dl, dlog_lbl, need_input_grad=True, groups=nl,
                            input_grad_out=dz[l0:l0 + nl].reshape(nl * B, D), input_grad_accumulate=True)
        del g
        if self.on_bucket_ready is not None:
            self.on_bucket_ready(0)      # decoder gradients are final
        # ---- PoE backward -> encoder heads
        if self.has_dropout:
            g_heads_img = torch.empty_like(heads_img)
            g_list = [g_heads_img[:B], g_heads_img[B:], torch.empty_like(heads_lbl)]
        else:
            g_list = [torch.empty_like(heads_img), torch.empty_like(heads_lbl)]
        K.poe_bwd(mus, lvs, self.masks_dev, self.noise, mu, lv, dz, None, None, self.coef[2],
                  [gg[:, :D] for gg in g_list], [gg[:, D:] for gg in g_list], m.POE_VARIANT,
                  dkl_per_term=True)
        # ---- encoders backward
        L.backward_tape(m.label_encoder.plan(), tape_lbl, g_list[-1])
        if self.has_dropout:
            d_hd = L.backward_tape(self.head, tape_head, g_heads_img, need_input_grad=True)
            d_h = torch.empty(B, d_hd.shape[1], dtype=torch.float32, device=self.dev)
            K.dropout_fanin_bwd(d_hd, self.drop_masks, d_h, 1.0 / KEEP)
            L.backward_tape(self.trunk, tape_trunk, d_h)
        else:
            L.backward_tape(m.image_encoder.plan(), tape_img, g_list[0])
        if self.on_bucket_ready is not None:
            self.on_bucket_ready(1)      # encoder gradients are final
        return elbo

    def step(self, image, label, annealing_factor, noise=None):
        """One eager step (no optimizer): zero_grad -> forward/backward.  Returns elbo[T+1]."""
        self.model.zero_grad(set_to_none=True)
        self.set_coefficients(annealing_factor)
        if noise is not None:
            self.set_noise(noise)
        else:
            self.draw_noise()
        return self.forward_backward(image, label)

    def terms_in_reference_order(self, elbo):
        """elbo[T+1] (engine order) -> [joint, image, label] + [total]."""
        idx = [self.ref_order.index(r) for r in range(self.T)] + [self.T]
        return elbo[idx]

    # ------------------------------------------------------------------ hipGraph capture
    def capture(self, optimizer, image_shape, label_example, warmup=3):
        """Record zero_grad + forward/backward + optimizer.step() into a hipGraph.  After this,
        ``replay(image, label, beta)`` copies the batch into static buffers, refreshes the loss
        coefficients and launches the graph: no per-kernel host work."""
        dev = self.dev
        self.static_image = torch.zeros((self.B,) + tuple(image_shape), dtype=torch.float32, device=dev)
        self.static_label = torch.zeros_like(label_example, device=dev)
        self.set_coefficients(1.0)
        snap = _snapshot(self.model, optimizer, self.counter)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._captured_body(optimizer)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._captured_body(optimizer)
        _restore(self.model, optimizer, self.counter, snap)   # warm-up steps must leave no trace
        torch.cuda.synchronize(dev)
        return self._graph

    def _captured_body(self, optimizer):
        self.model.zero_grad(set_to_none=True)
        self.draw_noise()
        self.forward_backward(self.static_image, self.static_label)
        optimizer.step()

    def replay(self, image, label, annealing_factor):
        self.static_image.copy_(image, non_blocking=True)
        self.static_label.copy_(label, non_blocking=True)
        self.set_coefficients(annealing_factor)
        self._graph.replay()
        return self.elbo

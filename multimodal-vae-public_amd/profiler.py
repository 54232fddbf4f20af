"""Per-launch timing with HIP events on the stream the kernels are launched on.

``with KernelProfile() as prof:`` wraps every launcher of ``kernels.py`` in a pair of
``torch.cuda.Event`` records on ``torch.cuda.current_stream()`` -- the same stream the C ABI
receives -- and tags it with the ALGORITHMIC work of the call (2*M*N*K flops for the GEMM-shaped
ops; for the HBM-bound ones the bytes SURVEY.md section 8(d) prescribes -- ``HBM_COSTS`` -- and the bytes of the
tensor arguments for the small plumbing kernels it does not list).  ``marker=True`` also launches an empty
``trace_marker_kernel`` in front of every call and keeps the call sequence (tools/step_by_shape.py).  ``bench.py`` uses it for the
``roofline`` object (live, in the same process as the timed run); the rocprofv3 kernel trace
committed under ``profiles/`` is the cross-check.
"""
import glob
import hashlib
import os

import torch

from . import kernels as K

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def code_stamp():
    """What a committed profile table was measured ON: sha256 (16 hex digits) over the kernel sources and the C header,
    plus the git head the collecting session was told about (``MVAE_GIT_HEAD``: the GPU box has no .git).  bench.py
    compares the hash with the tree it runs from and marks a table taken on other code ``profile_stale``."""
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, 'csrc', '*.hip')) + glob.glob(os.path.join(here, 'csrc', '*.h')))
    files.append(os.path.join(os.path.dirname(here), 'include', 'mvae_hip.h'))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return {'csrc_sha16': h.hexdigest()[:16], 'head': os.environ.get('MVAE_GIT_HEAD', 'unknown')}


def _numel_bytes(args, kwargs):
    n = 0
    for a in list(args) + list(kwargs.values()):
        if torch.is_tensor(a):
            n += a.numel() * a.element_size()
        elif isinstance(a, (list, tuple)):
            n += sum(t.numel() * t.element_size() for t in a if torch.is_tensor(t))
    return n


def _lin_cost(which):
    def cost(*a, **kw):
        if which == 'fwd':
            (M, Kd), N = a[0].shape, a[1].shape[0]
        elif which == 'dgrad':
            (M, N), Kd = a[0].shape, a[1].shape[1]
        else:
            (M, N), Kd = a[0].shape, a[1].shape[1]
        return 2.0 * M * N * Kd, 'M%d N%d K%d' % (M, N, Kd)
    return cost


def _conv_cost(kind):
    def cost(*a, **kw):
        # algorithmic flops = 2 * (elements of the conv-OUTPUT-shaped tensor) * Cin_conv * 16, where
        # "conv" is the plain convolution the op is a forward / dgrad / wgrad of
        if kind == 'conv2d_fwd':          # (x, w[Cout,Cin,4,4], pre, act)
            out = a[2] if a[2] is not None else a[3]
            fl = 2.0 * out.numel() * a[1].shape[1] * 16
            shape = tuple(out.shape)
        elif kind == 'convT2d_dgrad':     # (dy, w[CinT,CoutT,4,4], dx): a conv forward on dy producing dx
            fl = 2.0 * a[2].numel() * a[1].shape[1] * 16
            shape = tuple(a[2].shape)
        elif kind == 'conv2d_dgrad':
            dy, w = a[0], a[1]
            fl = 2.0 * dy.numel() * w.shape[1] * 16
            shape = tuple(a[2].shape)
        elif kind == 'convT2d_fwd':
            x, w = a[0], a[1]
            fl = 2.0 * x.numel() * w.shape[1] * 16
            shape = tuple((a[2] if a[2] is not None else a[3]).shape)
        elif kind == 'conv2d_wgrad':
            dy, x, dw = a[0], a[1], a[2]
            fl = 2.0 * dy.numel() * dw.shape[1] * 16
            shape = tuple(dw.shape)
        else:  # convT2d_wgrad: dy = grad of the transpose's output, x its input
            dy, x, dw = a[0], a[1], a[2]
            fl = 2.0 * x.numel() * dw.shape[1] * 16
            shape = tuple(dw.shape)
        return fl, 'x'.join(str(s) for s in shape)
    return cost


def _lin_grouped_cost(which):
    def cost(*a, **kw):
        if which == 'fwd':            # (x[G,M,K], w0[N,K], ...)
            (G, M, Kd), N = a[0].shape, a[1].shape[0]
        elif which == 'dgrad':        # (dy[G,M,N], w0[N,K], ...)
            (G, M, N), Kd = a[0].shape, a[1].shape[1]
        else:                         # (dy[G,M,N], x[G,M,K], ...)
            (G, M, N), Kd = a[0].shape, a[1].shape[2]
        return 2.0 * G * M * N * Kd, 'G%d M%d N%d K%d' % (G, M, N, Kd)
    return cost


def _wgrad_batched_cost(items, *a, **kw):
    fl = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _, _ in items)
    return fl, '%d layers' % len(items)


def _convT_stats_cost(x, w, stride, pad, *a, **kw):
    # the layer's flops; the shape names the output it would have stored
    B, Cin, H, W = x.shape
    OH, OW = (H - 1) * stride - 2 * pad + 4, (W - 1) * stride - 2 * pad + 4
    return 2.0 * x.numel() * w.shape[1] * 16, '%dx%dx%dx%d' % (B, w.shape[1], OH, OW)


GEMM_COSTS = {
    'convT2d_fwd_stats': _convT_stats_cost,
    'linear_wgrad_batched': _wgrad_batched_cost,
    'linear_fwd_grouped': _lin_grouped_cost('fwd'), 'linear_dgrad_grouped': _lin_grouped_cost('dgrad'),
    'linear_wgrad_grouped': _lin_grouped_cost('wgrad'),
    'linear_fwd': _lin_cost('fwd'), 'linear_dgrad': _lin_cost('dgrad'), 'linear_wgrad': _lin_cost('wgrad'),
    'linear_bce_fwd': _lin_cost('fwd'), 'linear_ce_fwd': _lin_cost('fwd'),      # (x, w, ...): the Linear's flops
    'conv2d_fwd': _conv_cost('conv2d_fwd'), 'conv2d_dgrad': _conv_cost('conv2d_dgrad'),
    'conv2d_wgrad': _conv_cost('conv2d_wgrad'), 'convT2d_fwd': _conv_cost('convT2d_fwd'),
    'convT2d_dgrad': _conv_cost('convT2d_dgrad'), 'convT2d_wgrad': _conv_cost('convT2d_wgrad'),
}
# ---- ALGORITHMIC bytes of the HBM-bound ops, SURVEY.md section 8(d) (what the OP must move, whatever kernel runs it):
#      Adam 28 B / parameter (read p, g, m, v; write p, m, v); BatchNorm forward 3 transfers of the activation (the
#      statistics need x before the apply reads it again: read, read, write), backward 5 (dy and x for the two sums,
#      dy and x again for the apply, dx written) -- the single-launch kernels of norm.hip move 2 / 3 for small slices
#      and can read above 1.0 of this figure's rate; BCE rows: read logits + target, write the row sums (+ d loss /
#      d logits when the launch also produces it); PoE + reparameterise + KL: read the experts' (mu, logvar) and eps,
#      write mu, logvar, z and the KL rows.
def _bn_fwd_bytes(x, gamma, beta, y, *a, **kw):
    return (3 if y is not None else 1) * x.numel() * 4


def _bn_bwd_bytes(dy, x, *a, **kw):
    return 5 * x.numel() * 4


def _adam_bytes(param, *a, **kw):
    return 28 * param.numel()


def _bce_rows_bytes(logits, target, rowsum, colw=None, drow=None, dlogits=None, *a, **kw):
    return 2 * logits.numel() * 4 + rowsum.numel() * 4 + (logits.numel() * 4 if dlogits is not None else 0)


def _poe_fwd_bytes(mus, lvs, masks_dev, noise, mu, logvar, z, kl, *a, **kw):
    T, B, D = mu.shape
    E = len(mus)
    return (2 * E + (T if noise is not None else 0)) * B * D * 4 + 3 * T * B * D * 4 + T * B * 4


def _poe_draw_bytes(mus, lvs, masks_dev, noise_out, seed, counter_dev, counter_offset, mu, logvar, z, kl, *a, **kw):
    T, B, D = mu.shape
    return 2 * len(mus) * B * D * 4 + 4 * T * B * D * 4 + T * B * 4       # eps is written, not read


def _poe_bwd_bytes(mus, lvs, masks_dev, noise, mu, logvar, *a, **kw):
    T, B, D = mu.shape
    E = len(mus)
    # read the experts, eps, the fused (mu, logvar) and dz per term; write the experts' gradients
    return (2 * E + 4 * T) * B * D * 4 + 2 * E * B * D * 4


HBM_COSTS = {
    'bn_train_fwd': _bn_fwd_bytes, 'bn_train_bwd': _bn_bwd_bytes,
    'adam_step': _adam_bytes, 'adam_apply': _adam_bytes, 'adam_apply_at': _adam_bytes,
    'bce_rowsum_fwd': _bce_rows_bytes,
    'poe_fwd': _poe_fwd_bytes, 'poe_fwd_draw': _poe_draw_bytes, 'poe_bwd': _poe_bwd_bytes, 'poe_bwd_split': _poe_bwd_bytes,
}

HBM_OPS = ['bn_train_fwd', 'bn_train_bwd', 'bn_eval_fwd', 'swish_fwd', 'swish_bwd', 'embedding_swish_fwd',
           'embedding_swish_bwd', 'poe_fwd', 'poe_bwd', 'kl_rows_fwd', 'kl_rows_bwd', 'bce_rowsum_fwd',
           'bce_rowsum_bwd', 'ce_fwd', 'ce_bwd', 'group_sums', 'randn_', 'bernoulli_', 'adam_step', 'fill_',
           'dropout_fanout_fwd', 'dropout_fanin_bwd', 'bce_elem_fwd', 'bce_elem_bwd', 'embedding_swish_fwd_grouped',
           'embedding_swish_bwd_grouped', 'block_gather', 'block_scatter_add', 'elbo_reduce', 'philox_fill',
           'adam_apply', 'sigmoid_fwd', 'affine_fwd', 'scatter_sums', 'poe_fwd_draw', 'poe_bwd_split', 'adam_apply_at',
           'counter_add', 'conv_repack_batched', 'bn_stats_merge']


class KernelProfile(object):
    def __init__(self, marker=False, timed=True):
        self.marker = marker                 # an empty kernel in front of every call + the call sequence
        self.timed = timed                   # False: no event pairs (a rocprofv3 trace supplies the durations)
        self.sequence = []                   # [(name, key)] in launch order
        self.records = []
        self.last_call = {}      # (name, key) -> (launcher, args, kwargs) of the most recent call
        self._saved = {}

    def _wrap(self, name, fn, cost):
        def wrapped(*a, **kw):
            if self.marker:
                K.trace_marker(len(self.sequence))
            e0 = e1 = None
            if self.timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            out = fn(*a, **kw)
            if self.timed:
                e1.record()
            if cost is not None:
                flops, key = cost(*a, **kw)
                nbytes = 0
            else:
                by = HBM_COSTS.get(name)
                flops, key, nbytes = 0.0, '', (by(*a, **kw) if by is not None else _numel_bytes(a, kw))
                if by is not None:       # the listed ops are reported per shape (a step has BatchNorms of 14 sizes)
                    first = next((t for t in a if torch.is_tensor(t)), None)
                    key = 'x'.join(str(s) for s in first.shape) if first is not None else '%d B' % nbytes
            self.sequence.append((name, key, flops, nbytes))
            if not self.timed:
                return out
            self.records.append((name, key, flops, nbytes, e0, e1))
            self.last_call[(name, key)] = (fn, a, kw)
            return out
        return wrapped

    def __enter__(self):
        for name in list(GEMM_COSTS) + HBM_OPS:
            fn = getattr(K, name)
            self._saved[name] = fn
            setattr(K, name, self._wrap(name, fn, GEMM_COSTS.get(name)))
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(K, name, fn)
        self._saved = {}
        return False

    def steady_state_ms(self, name, key, launches=20, replays=5):
        """Average duration of ONE launch of the recorded call (name, key) with the GPU queue kept
        full: ``launches`` copies captured into a hipGraph on a private stream, replayed
        ``replays`` times between two HIP events on that stream.  The per-call event pairs of the
        eager pass include host enqueue gaps (python is slower than a 10 us kernel); this does not."""
        fn, a, kw = self.last_call[(name, key)]
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn(*a, **kw)                     # scratch of this stream exists before capture
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(launches):
                fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            g.replay()
            e0.record()
            for _ in range(replays):
                g.replay()
            e1.record()
        st.synchronize()
        torch.cuda.current_stream().wait_stream(st)
        return e0.elapsed_time(e1) / (launches * replays)

    def summary(self):
        """[{name, key, calls, ms_total, ms_avg, flops, bytes, tflops, gbs}] sorted by total time."""
        torch.cuda.synchronize()
        agg = {}
        for name, key, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault((name, key), [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += flops
            a[3] += nbytes
        rows = []
        for (name, key), (calls, ms, flops, nbytes) in agg.items():
            rows.append(dict(name=name, key=key, calls=calls, ms_total=ms, ms_avg=ms / calls,
                             flops=flops / calls, bytes=nbytes / calls,
                             tflops=(flops / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                             gbs=(nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0))
        rows.sort(key=lambda r: -r['ms_total'])
        return rows

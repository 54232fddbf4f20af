"""Tensor-level launchers over the C ABI (one python function per ``mvae_*`` entry point).

Everything here is enqueue-only on ``torch.cuda.current_stream()``: no host sync, no
allocation other than the cached scratch buffer, so a whole train step can be captured into
a hipGraph (``torch.cuda.CUDAGraph``).  PyTorch is used for device memory and streams only.
Tensors must live on the GPU, be fp32 (labels int64) and contiguous unless noted.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACCUMULATE, ACT_SWISH, check

_WS = {}          # device index -> list of scratch tensors (old ones kept alive for captured graphs)
_WS_MIN_BYTES = 64 << 20


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('multimodal-vae-public_amd: the HIP path needs GPU tensors (got %s); '
                               'there is no CPU fallback' % t.device)


def _f32c(*tensors):
    for t in tensors:
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError('expected a contiguous float32 tensor, got %s %s contiguous=%s'
                               % (t.dtype, tuple(t.shape), t.is_contiguous()))


def workspace(nbytes, device):
    """Scratch for split reductions.  Grows by allocating a new buffer; earlier buffers stay
    alive because captured graphs may hold their addresses."""
    # one scratch per (device, stream): engine branches on different streams run concurrently
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    bufs = _WS.setdefault(key, [])
    if not bufs or bufs[-1].numel() * 4 < nbytes:
        n = max(int(nbytes), _WS_MIN_BYTES)
        bufs.append(torch.empty((n + 3) // 4, dtype=torch.float32, device=device))
    return bufs[-1]


def _ws_args(nbytes, device):
    ws = workspace(nbytes, device)
    return _ptr(ws), ctypes.c_size_t(ws.numel() * 4)


# ---------------------------------------------------------------------------- Linear
def linear_fwd(x, w, bias, pre=None, act=None, mask=None, mask_scale=1.0, ldx=None, ldy=None):
    """pre[M,N] = x.w^T + b ; act = swish(pre) * mask*scale.  x may be a row-strided view (ldx)."""
    _need_gpu(x, w, bias, pre, act, mask)
    M, K = x.shape
    N = w.shape[0]
    ldx = ldx if ldx is not None else x.stride(0)
    out = pre if pre is not None else act
    ldy = ldy if ldy is not None else out.stride(0)
    ws, wsb = _ws_args(_lib.lib().mvae_gemm_ws_bytes(M, N, K), x.device)
    check(_lib.lib().mvae_linear_fwd(_ptr(x), ldx, _ptr(w), _ptr(bias), _ptr(pre), _ptr(act), ldy,
                                     _ptr(mask), mask_scale, M, N, K, ws, wsb, _stream()), 'mvae_linear_fwd')


def bce_partials(N):
    """Partial sums per row that ``linear_bce_fwd`` writes for N columns."""
    return (N + 31) // 32


def linear_bce_fwd(x, w, bias, target, drow, dlogits, partial, rows_per_group, target_rows, logits=None):
    """Last Linear of a decoder + its Bernoulli term: dlogits[M,N] and partial[M, bce_partials(N)] (a row's
    term = the sum of its partials); the logits are stored only when ``logits`` is given."""
    _need_gpu(x, w, bias, target, drow, dlogits, partial, logits)
    M, K = x.shape
    N = w.shape[0]
    if target.dim() != 2 or target.stride(1) != 1 or target.shape[1] != N or target.shape[0] < target_rows:
        raise RuntimeError('target must be [>= target_rows, N] with unit column stride')
    if partial.numel() < M * bce_partials(N) or not partial.is_contiguous():
        raise RuntimeError('partial must hold M * ceil(N / 32) floats')
    if logits is not None and logits.stride() != dlogits.stride():
        raise RuntimeError('logits and dlogits must share a layout')
    check(_lib.lib().mvae_linear_bce_fwd(_ptr(x), x.stride(0), _ptr(w), _ptr(bias), _ptr(target), target_rows,
                                         target.stride(0), _ptr(drow), rows_per_group, _ptr(dlogits),
                                         dlogits.stride(0), _ptr(logits), _ptr(partial), M, N, K, _stream()),
          'mvae_linear_bce_fwd')


def linear_ce_fwd(x, w, bias, label, drow, dlogits, row, rows_per_group, label_rows, logits=None):
    """Last Linear of a decoder (N <= 32 classes) + its categorical term: row[M] and dlogits[M,N]."""
    _need_gpu(x, w, bias, label, drow, dlogits, row, logits)
    M, K = x.shape
    N = w.shape[0]
    if label.dtype != torch.int64 or not label.is_contiguous() or label.numel() < label_rows:
        raise RuntimeError('label must be a contiguous int64 vector of at least label_rows entries')
    if logits is not None and logits.stride() != dlogits.stride():
        raise RuntimeError('logits and dlogits must share a layout')
    check(_lib.lib().mvae_linear_ce_fwd(_ptr(x), x.stride(0), _ptr(w), _ptr(bias), _ptr(label), label_rows,
                                        _ptr(drow), rows_per_group, _ptr(dlogits), dlogits.stride(0), _ptr(logits),
                                        _ptr(row), M, N, K, _stream()), 'mvae_linear_ce_fwd')


def linear_dgrad(dy, w, dx, pre_in=None, mask=None, mask_scale=1.0, accumulate=False):
    _need_gpu(dy, w, dx, pre_in, mask)
    M, N = dy.shape
    K = w.shape[1]
    ws, wsb = _ws_args(_lib.lib().mvae_gemm_ws_bytes(M, K, N), dy.device)
    check(_lib.lib().mvae_linear_dgrad(_ptr(dy), dy.stride(0), _ptr(w), _ptr(dx), dx.stride(0),
                                       _ptr(pre_in), _ptr(mask), mask_scale, M, N, K,
                                       ACCUMULATE if accumulate else 0, ws, wsb, _stream()), 'mvae_linear_dgrad')


def linear_wgrad(dy, x, dw, db=None, accumulate=False):
    _need_gpu(dy, x, dw, db)
    M, N = dy.shape
    K = x.shape[1]
    nbytes = _lib.lib().mvae_gemm_ws_bytes(N, K, M)
    ws, wsb = _ws_args(nbytes, dy.device)
    check(_lib.lib().mvae_linear_wgrad(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), _ptr(db),
                                       M, N, K, ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_linear_wgrad')


def wgrad_batchable(dy, x):
    """Whether mvae_linear_wgrad_batched takes this problem (see include/mvae_hip.h)."""
    M, N = dy.shape
    K = x.shape[1]
    tiles = ((N + 31) // 32) * ((K + 31) // 32)
    return (tiles <= 2048 and M <= 4096 and M * dy.stride(0) * 4 < (1 << 32) and M * x.stride(0) * 4 < (1 << 32)
            and dy.stride(1) == 1 and x.stride(1) == 1)


def linear_wgrad_batched(items, adam=None):
    """items: [(dy[M,N], x[M,K], dw[N,K], db[N] or None, accumulate)] -- the weight gradients of several Linear
    layers in ONE launch (mvae_linear_wgrad_batched); more than WGRAD_BATCH_MAX items go out in several.
    ``adam`` (an ``_lib.AdamFuse``, see ``optim.FusedAdam.fuse_record``): the launch also applies Adam to the
    parameters behind these gradients (mvae_linear_wgrad_batched_adam); an item (None, None, g, None, False) is a
    finished flat gradient that only takes the update."""
    items = list(items)
    for lo in range(0, len(items), _lib.WGRAD_BATCH_MAX):
        chunk = items[lo:lo + _lib.WGRAD_BATCH_MAX]
        arr = (_lib.WgradItem * len(chunk))()
        for q, (dy, x, dw, db, acc) in enumerate(chunk):
            if dy is None:
                if adam is None:
                    raise RuntimeError('update-only items need adam=')
                _need_gpu(dw); _f32c(dw)
                arr[q] = _lib.WgradItem(None, 0, None, 0, _ptr(dw), None, 0, 0, dw.numel(), 0)
                continue
            _need_gpu(dy, x, dw, db); _f32c(dw, db)
            M, N = dy.shape
            if x.shape[0] != M or tuple(dw.shape) != (N, x.shape[1]):
                raise RuntimeError('wgrad item %d: shapes %s %s %s' % (q, tuple(dy.shape), tuple(x.shape), tuple(dw.shape)))
            arr[q] = _lib.WgradItem(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), _ptr(db), M, N,
                                    x.shape[1], ACCUMULATE if acc else 0)
        if adam is not None:
            check(_lib.lib().mvae_linear_wgrad_batched_adam(arr, len(chunk), ctypes.byref(adam), _stream()),
                  'mvae_linear_wgrad_batched_adam')
        else:
            check(_lib.lib().mvae_linear_wgrad_batched(arr, len(chunk), _stream()), 'mvae_linear_wgrad_batched')


# ---------------------------------------------------------------------------- grouped Linear
# G problems of one shape per launch: activations are [G, rows, width] tensors, parameters are
# (tensor of group 0, stride in floats to the same tensor of the next group) -- the experts' slices
# of the parameter arena.
def _g3(t, name):
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) < t.shape[1] * t.stride(1):
        raise RuntimeError('%s must be a [groups, rows, width] tensor with unit column stride' % name)
    return t


def linear_fwd_grouped(x, w0, w_gs, b0, b_gs, pre=None, act=None):
    _need_gpu(x, w0, b0, pre, act)
    G, M, K = _g3(x, 'x').shape
    N = w0.shape[0]
    out = _g3(pre if pre is not None else act, 'output')
    if pre is not None and act is not None and (act.stride() != pre.stride()):
        raise RuntimeError('pre and act must share a layout')
    ws, wsb = _ws_args(G * _lib.lib().mvae_gemm_ws_bytes(M, N, K), x.device)
    check(_lib.lib().mvae_linear_fwd_grouped(_ptr(x), x.stride(1), x.stride(0), _ptr(w0), w_gs, _ptr(b0), b_gs,
                                             _ptr(pre), _ptr(act), out.stride(1), out.stride(0), G, M, N, K,
                                             ws, wsb, _stream()), 'mvae_linear_fwd_grouped')


def linear_dgrad_grouped(dy, w0, w_gs, dx, pre_in=None, accumulate=False):
    _need_gpu(dy, w0, dx, pre_in)
    G, M, N = _g3(dy, 'dy').shape
    K = w0.shape[1]
    _g3(dx, 'dx')
    if pre_in is not None and (not pre_in.is_contiguous() or pre_in.shape != (G, M, K)):
        raise RuntimeError('pre_in must be a contiguous [G, M, K] tensor')
    ws, wsb = _ws_args(G * _lib.lib().mvae_gemm_ws_bytes(M, K, N), dy.device)
    check(_lib.lib().mvae_linear_dgrad_grouped(_ptr(dy), dy.stride(1), dy.stride(0), _ptr(w0), w_gs, _ptr(dx),
                                               dx.stride(1), dx.stride(0), _ptr(pre_in), M * K, G, M, N, K,
                                               ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_linear_dgrad_grouped')


def linear_wgrad_grouped(dy, x, dw0, dw_gs, db0=None, db_gs=0, accumulate=False):
    _need_gpu(dy, x, dw0, db0)
    G, M, N = _g3(dy, 'dy').shape
    K = _g3(x, 'x').shape[2]
    ws, wsb = _ws_args(G * _lib.lib().mvae_gemm_ws_bytes(N, K, M), dy.device)
    check(_lib.lib().mvae_linear_wgrad_grouped(_ptr(dy), dy.stride(1), dy.stride(0), _ptr(x), x.stride(1),
                                               x.stride(0), _ptr(dw0), dw_gs, _ptr(db0), db_gs, G, M, N, K,
                                               ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_linear_wgrad_grouped')


# ---- pairs: two same-shaped Linear problems anywhere in memory as ONE grouped launch (G = 2); the group
#      stride is the pointer difference modulo 2^64.  MNIST's image and text decoders share their first
#      three layer shapes (mnist/model.py:96-104,137-145): pairing them halves those launches.
_U64 = (1 << 64) - 1


def _pair_stride(a, what):
    a0, a1 = a
    if (a0 is None) != (a1 is None):
        raise RuntimeError('%s: both or neither of a pair' % what)
    if a0 is None:
        return 0
    if a0.shape != a1.shape or a0.stride() != a1.stride() or a0.dtype != torch.float32 or a1.dtype != torch.float32:
        raise RuntimeError('%s: the two problems of a pair must share shape, strides and dtype' % what)
    d = a1.data_ptr() - a0.data_ptr()
    if d % 4:
        raise RuntimeError('%s: misaligned pair' % what)
    return (d // 4) & _U64


def _rows2d(t, what):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError('%s must be [rows, width] with unit column stride' % what)
    return t


def linear_fwd_pair(x, w, b, pre, act):
    """(x0, x1), (w0, w1), (b0, b1), (pre0, pre1), (act0, act1): pre_g = x_g.w_g^T + b_g, act = swish."""
    _need_gpu(*x, *w, *b, *pre, *act)
    M, K = _rows2d(x[0], 'x').shape
    N = w[0].shape[0]
    out = pre if pre[0] is not None else act
    if pre[0] is not None and act[0] is not None and pre[0].stride() != act[0].stride():
        raise RuntimeError('pre and act must share a layout')
    if _pair_stride(pre, 'pre') != _pair_stride(act, 'act') and pre[0] is not None and act[0] is not None:
        raise RuntimeError('pre and act pairs must be laid out alike')
    ws, wsb = _ws_args(2 * _lib.lib().mvae_gemm_ws_bytes(M, N, K), x[0].device)
    check(_lib.lib().mvae_linear_fwd_grouped(_ptr(x[0]), x[0].stride(0), _pair_stride(x, 'x'), _ptr(w[0]),
                                             _pair_stride(w, 'w'), _ptr(b[0]), _pair_stride(b, 'bias'),
                                             _ptr(pre[0]), _ptr(act[0]), _rows2d(out[0], 'out').stride(0),
                                             _pair_stride(out, 'out'), 2, M, N, K, ws, wsb, _stream()),
          'mvae_linear_fwd_grouped')


def linear_dgrad_pair(dy, w, dx, pre_in=(None, None), accumulate=False):
    _need_gpu(*dy, *w, *dx, *pre_in)
    M, N = _rows2d(dy[0], 'dy').shape
    K = w[0].shape[1]
    if pre_in[0] is not None and not (pre_in[0].is_contiguous() and pre_in[0].shape == (M, K)):
        raise RuntimeError('pre_in must be contiguous [M, K]')
    ws, wsb = _ws_args(2 * _lib.lib().mvae_gemm_ws_bytes(M, K, N), dy[0].device)
    check(_lib.lib().mvae_linear_dgrad_grouped(_ptr(dy[0]), dy[0].stride(0), _pair_stride(dy, 'dy'), _ptr(w[0]),
                                               _pair_stride(w, 'w'), _ptr(dx[0]), _rows2d(dx[0], 'dx').stride(0),
                                               _pair_stride(dx, 'dx'), _ptr(pre_in[0]), _pair_stride(pre_in, 'pre_in'),
                                               2, M, N, K, ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_linear_dgrad_grouped')


def linear_wgrad_pair(dy, x, dw, db=(None, None), accumulate=False):
    _need_gpu(*dy, *x, *dw, *db)
    M, N = _rows2d(dy[0], 'dy').shape
    K = _rows2d(x[0], 'x').shape[1]
    ws, wsb = _ws_args(2 * _lib.lib().mvae_gemm_ws_bytes(N, K, M), dy[0].device)
    check(_lib.lib().mvae_linear_wgrad_grouped(_ptr(dy[0]), dy[0].stride(0), _pair_stride(dy, 'dy'), _ptr(x[0]),
                                               x[0].stride(0), _pair_stride(x, 'x'), _ptr(dw[0]),
                                               _pair_stride(dw, 'dw'), _ptr(db[0]), _pair_stride(db, 'db'), 2, M, N, K,
                                               ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_linear_wgrad_grouped')


def embedding_swish_fwd_grouped(idx, w0, w_gs, act):
    """idx: float [R, G] ({0,1} columns, celeba19's attrs); act [G, R, width] contiguous."""
    _need_gpu(idx, w0, act)
    if idx.dtype != torch.float32 or idx.dim() != 2 or idx.stride(1) != 1 or not act.is_contiguous():
        raise RuntimeError('grouped embedding wants a float [rows, groups] index and a contiguous output')
    R, G = idx.shape
    check(_lib.lib().mvae_embedding_swish_fwd_grouped(_ptr(idx), idx.stride(0), 1, _ptr(w0), w_gs, _ptr(act),
                                                      act.stride(0), G, R, w0.shape[0], w0.shape[1], _stream()),
          'mvae_embedding_swish_fwd_grouped')


def embedding_swish_bwd_grouped(idx, w0, w_gs, dact, dw0, accumulate=False):
    _need_gpu(idx, w0, dact, dw0)
    if idx.dtype != torch.float32 or idx.dim() != 2 or idx.stride(1) != 1 or not dact.is_contiguous():
        raise RuntimeError('grouped embedding wants a float [rows, groups] index and a contiguous gradient')
    R, G = idx.shape
    check(_lib.lib().mvae_embedding_swish_bwd_grouped(_ptr(idx), idx.stride(0), 1, _ptr(w0), w_gs, _ptr(dact),
                                                      dact.stride(0), _ptr(dw0), G, R, w0.shape[0], w0.shape[1],
                                                      ACCUMULATE if accumulate else 0, _stream()),
          'mvae_embedding_swish_bwd_grouped')


# ---------------------------------------------------------------------------- Conv 4x4
def _conv_call(name, a, b, c, d, B, Cin, H, W, Cout, stride, pad, repack=False, wr=None):
    if repack and wr is not None:       # the caller made the repacked copy already (conv_repack_batched): w = NULL
        _need_gpu(wr); _f32c(wr)
        check(getattr(_lib.lib(), name)(_ptr(a), None, _ptr(c), _ptr(d), B, Cin, H, W, Cout, stride, pad,
                                        _ptr(wr), wr.numel() * 4, _stream()), name)
    elif repack:    # dgrad-form launches repack the weights (Cout*Cin*16 floats) into scratch first
        ws, wsb = _ws_args(Cout * Cin * 16 * 4, a.device)
        check(getattr(_lib.lib(), name)(_ptr(a), _ptr(b), _ptr(c), _ptr(d), B, Cin, H, W, Cout, stride, pad,
                                        ws, wsb, _stream()), name)
    else:
        check(getattr(_lib.lib(), name)(_ptr(a), _ptr(b), _ptr(c), _ptr(d), B, Cin, H, W, Cout, stride, pad,
                                        _stream()), name)


def conv2d_fwd(x, w, pre, act, stride, pad):
    _need_gpu(x, w, pre, act); _f32c(x, w, pre, act)
    B, Cin, H, W = x.shape
    _conv_call('mvae_conv2d_k4_fwd', x, w, pre, act, B, Cin, H, W, w.shape[0], stride, pad)


def conv2d_dgrad(dy, w, dx, pre_in, stride, pad, wr=None):
    _need_gpu(dy, w, dx, pre_in); _f32c(dy, w, dx, pre_in)
    B, Cin, H, W = dx.shape
    _conv_call('mvae_conv2d_k4_dgrad', dy, w, dx, pre_in, B, Cin, H, W, w.shape[0], stride, pad, repack=True, wr=wr)


def conv_repack_floats(transposed, w, B, Cin, H, W, Cout, stride, pad):
    """Floats of the repacked weight copy a Conv2d data gradient (transposed False) / ConvTranspose2d forward
    (True) of this geometry reads, or 0 when that launch reads ``w`` itself (include/mvae_hip.h)."""
    return int(_lib.lib().mvae_conv_k4_repack_floats(1 if transposed else 0, _ptr(w), B, Cin, H, W, Cout, stride, pad))


def conv_repack_batched(items):
    """items: [(w, wr, transposed, Cin, Cout, stride, pad)] -- all repacked copies in one launch (16 per launch)."""
    items = list(items)
    for lo in range(0, len(items), _lib.REPACK_MAX):
        chunk = items[lo:lo + _lib.REPACK_MAX]
        arr = (_lib.RepackItem * len(chunk))()
        for q, (w, wr, tr, Cin, Cout, s, p) in enumerate(chunk):
            _need_gpu(w, wr); _f32c(w, wr)
            if wr.numel() < Cin * Cout * 16:
                raise RuntimeError('repack buffer too small')
            arr[q] = _lib.RepackItem(_ptr(w), _ptr(wr), 1 if tr else 0, Cin, Cout, s, p)
        check(_lib.lib().mvae_conv_k4_repack_batched(arr, len(chunk), _stream()), 'mvae_conv_k4_repack_batched')


def conv2d_wgrad(dy, x, dw, stride, pad, accumulate=False):
    _need_gpu(dy, x, dw); _f32c(dy, x, dw)
    B, Cin, H, W = x.shape
    Cout = dw.shape[0]
    nbytes = _lib.lib().mvae_gemm_ws_bytes(Cout, Cin * 16, B * dy.shape[2] * dy.shape[3])
    ws, wsb = _ws_args(nbytes, dy.device)
    check(_lib.lib().mvae_conv2d_k4_wgrad(_ptr(dy), _ptr(x), _ptr(dw), B, Cin, H, W, Cout, stride, pad,
                                          ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_conv2d_k4_wgrad')


def convT2d_fwd(x, w, pre, act, stride, pad, wr=None):
    """x[B,Cin,H,W], w[Cin,Cout,4,4] -> [B,Cout,(H-1)s-2p+4, ...]"""
    _need_gpu(x, w, pre, act); _f32c(x, w, pre, act)
    B, Cin, H, W = x.shape
    _conv_call('mvae_convT2d_k4_fwd', x, w, pre, act, B, Cin, H, W, w.shape[1], stride, pad, repack=True, wr=wr)


def convT2d_stats_tiles(x, w, stride, pad):
    """Records the statistics-only launch would write for ConvTranspose2d(w)(x), 0 if the shape is not covered."""
    B, Cin, H, W = x.shape
    return int(_lib.lib().mvae_convT2d_k4_stats_tiles(B, Cin, H, W, w.shape[1], stride, pad))


def convT2d_fwd_stats(x, w, stride, pad, wr=None):
    """ConvTranspose2d(w)(x) WITHOUT storing it: returns part [tiles, Cout, 2] = (mean, M2) of every 512-element
    column tile per output channel (mvae_convT2d_k4_fwd_stats), for ``bn_stats_merge``.  ``wr``: the repacked
    weight copy made ahead (conv_repack_batched)."""
    _need_gpu(x, w, wr); _f32c(x, w, wr)
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    tiles = convT2d_stats_tiles(x, w, stride, pad)
    if tiles <= 0:
        raise RuntimeError('statistics-only transposed conv: shape %s -> %d channels not covered' % (tuple(x.shape), Cout))
    part = torch.empty(tiles, Cout, 2, dtype=torch.float32, device=x.device)
    if wr is not None:
        ws, wsb, wp = _ptr(wr), wr.numel() * 4, None
    else:
        ws, wsb = _ws_args(Cin * Cout * 16 * 4, x.device)
        wp = _ptr(w)
    check(_lib.lib().mvae_convT2d_k4_fwd_stats(_ptr(x), wp, _ptr(part), part.numel(), B, Cin, H, W, Cout, stride, pad,
                                               ws, wsb, _stream()), 'mvae_convT2d_k4_fwd_stats')
    return part


def bn_stats_merge(part, G, save_mean, save_invstd, running_mean, running_var, eps=1e-5, momentum=0.1, n_updates=1,
                   n_updates_dev=None):
    """Saved + running BatchNorm statistics of G groups from the records of ``convT2d_fwd_stats``."""
    _need_gpu(part, save_mean, save_invstd, running_mean, running_var)
    _f32c(part, save_mean, save_invstd, running_mean, running_var)
    tiles, C = part.shape[0], part.shape[1]
    check(_lib.lib().mvae_bn_stats_merge(_ptr(part), tiles, _lib.STATS_TILE_ELEMS, G, C, _ptr(save_mean), _ptr(save_invstd),
                                         _ptr(running_mean), _ptr(running_var), eps, momentum, n_updates,
                                         _ptr(n_updates_dev), _stream()), 'mvae_bn_stats_merge')


def convT2d_dgrad(dy, w, dx, pre_in, stride, pad):
    _need_gpu(dy, w, dx, pre_in); _f32c(dy, w, dx, pre_in)
    B, Cin, H, W = dx.shape
    _conv_call('mvae_convT2d_k4_dgrad', dy, w, dx, pre_in, B, Cin, H, W, w.shape[1], stride, pad)


def convT2d_wgrad(dy, x, dw, stride, pad, accumulate=False):
    _need_gpu(dy, x, dw); _f32c(dy, x, dw)
    B, Cin, H, W = x.shape
    Cout = dw.shape[1]
    nbytes = _lib.lib().mvae_gemm_ws_bytes(Cin, Cout * 16, B * H * W)
    ws, wsb = _ws_args(nbytes, dy.device)
    check(_lib.lib().mvae_convT2d_k4_wgrad(_ptr(dy), _ptr(x), _ptr(dw), B, Cin, H, W, Cout, stride, pad,
                                           ACCUMULATE if accumulate else 0, ws, wsb, _stream()),
          'mvae_convT2d_k4_wgrad')


# ---------------------------------------------------------------------------- BatchNorm
def bn_train_fwd(x, gamma, beta, y, save_mean, save_invstd, running_mean, running_var, G,
                 eps=1e-5, momentum=0.1, n_updates=1, swish=True, n_updates_dev=None):
    """x [G*B, C, *spatial]; save_* [G, C].  ``n_updates_dev``: device int32[1] overriding
    ``n_updates`` (celeba19: how many terms of this step contain the image)."""
    _need_gpu(x, gamma, beta, y, save_mean, save_invstd, running_mean, running_var)
    _f32c(x, gamma, beta, y, save_mean, save_invstd, running_mean, running_var)
    GB, C = x.shape[0], x.shape[1]
    HW = x.numel() // (GB * C)
    B = GB // G
    nbytes = _lib.lib().mvae_bn_ws_bytes(G, C, B * HW)
    ws, wsb = _ws_args(nbytes, x.device)
    check(_lib.lib().mvae_bn_train_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(save_mean),
                                       _ptr(save_invstd), _ptr(running_mean), _ptr(running_var),
                                       G, B, C, HW, eps, momentum, n_updates, _ptr(n_updates_dev),
                                       ACT_SWISH if swish else 0,
                                       ws, wsb, _stream()), 'mvae_bn_train_fwd')


def bn_train_bwd(dy, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, G, swish=True,
                 accumulate=False):
    _need_gpu(dy, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta)
    _f32c(dy, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta)
    GB, C = x.shape[0], x.shape[1]
    HW = x.numel() // (GB * C)
    B = GB // G
    nbytes = _lib.lib().mvae_bn_ws_bytes(G, C, B * HW)
    ws, wsb = _ws_args(nbytes, x.device)
    flags = (ACT_SWISH if swish else 0) | (ACCUMULATE if accumulate else 0)
    check(_lib.lib().mvae_bn_train_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                       _ptr(save_invstd), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                       G, B, C, HW, flags, ws, wsb, _stream()), 'mvae_bn_train_bwd')


def bn_eval_fwd(x, gamma, beta, y, running_mean, running_var, eps=1e-5, swish=True):
    _need_gpu(x, gamma, beta, y, running_mean, running_var)
    _f32c(x, gamma, beta, y, running_mean, running_var)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    check(_lib.lib().mvae_bn_eval_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(running_mean),
                                      _ptr(running_var), N, C, HW, eps, ACT_SWISH if swish else 0,
                                      _stream()), 'mvae_bn_eval_fwd')


# ---------------------------------------------------------------------------- elementwise / embedding
def reparam_fwd(mu, logvar, eps, z):
    """z = eps * exp(logvar / 2) + mu (mnist/model.py:29-35), elementwise."""
    _need_gpu(mu, logvar, eps, z); _f32c(mu, logvar, eps, z)
    check(_lib.lib().mvae_reparam_fwd(_ptr(mu), _ptr(logvar), _ptr(eps), _ptr(z), mu.numel(), _stream()),
          'mvae_reparam_fwd')


def sigmoid_fwd(x, y):
    _need_gpu(x, y); _f32c(x, y)
    check(_lib.lib().mvae_sigmoid_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), 'mvae_sigmoid_fwd')


def affine_fwd(x, scale, shift, y):
    """y = x * scale + shift with scale / shift of ``period`` = scale.numel() elements repeated along x."""
    _need_gpu(x, scale, shift, y); _f32c(x, scale, shift, y)
    if scale.numel() != shift.numel() or x.numel() % scale.numel():
        raise RuntimeError('affine_fwd: scale/shift must have equal sizes dividing x')
    check(_lib.lib().mvae_affine_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), x.numel(), scale.numel(),
                                     _stream()), 'mvae_affine_fwd')


def swish_fwd(x, y):
    _need_gpu(x, y); _f32c(x, y)
    check(_lib.lib().mvae_swish_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), 'mvae_swish_fwd')


def swish_bwd(dy, x, dx):
    _need_gpu(dy, x, dx); _f32c(dy, x, dx)
    check(_lib.lib().mvae_swish_bwd(_ptr(dy), _ptr(x), _ptr(dx), x.numel(), _stream()), 'mvae_swish_bwd')


def _index_kind(idx):
    """0 = contiguous int64 labels; s > 0 = fp32 {0,1} values with element stride s (a column of
    attrs[B,18] has s = 18)."""
    if idx.dim() != 1:
        raise RuntimeError('embedding index must be 1-D')
    if idx.dtype == torch.int64:
        if idx.numel() > 1 and idx.stride(0) != 1:
            raise RuntimeError('int64 embedding index must be contiguous')
        return 0
    if idx.dtype == torch.float32:
        return max(int(idx.stride(0)), 1)
    raise RuntimeError('embedding index must be int64 or float32, got %s' % idx.dtype)


def block_gather(src, idx_dev, dst, block_elems):
    _need_gpu(src, idx_dev, dst); _f32c(src, dst)
    check(_lib.lib().mvae_block_gather(_ptr(src), _ptr(idx_dev), _ptr(dst), idx_dev.numel(), block_elems,
                                       _stream()), 'mvae_block_gather')


def block_scatter_add(src, idx_dev, dst, n_dst, block_elems):
    _need_gpu(src, idx_dev, dst); _f32c(src, dst)
    check(_lib.lib().mvae_block_scatter_add(_ptr(src), _ptr(idx_dev), _ptr(dst), idx_dev.numel(), n_dst,
                                            block_elems, _stream()), 'mvae_block_scatter_add')


def scatter_sums(vals, coef, idx_dev, out, total, accumulate_total=True):
    _need_gpu(vals, coef, idx_dev, out, total); _f32c(vals, coef, out, total)
    check(_lib.lib().mvae_scatter_sums(_ptr(vals), _ptr(coef), _ptr(idx_dev), _ptr(out), _ptr(total),
                                       vals.numel(), ACCUMULATE if accumulate_total else 0, _stream()),
          'mvae_scatter_sums')


def embedding_swish_fwd(idx, w, act):
    _need_gpu(idx, w, act); _f32c(w, act)
    check(_lib.lib().mvae_embedding_swish_fwd(_ptr(idx), _index_kind(idx), _ptr(w), _ptr(act), idx.numel(),
                                              w.shape[0], w.shape[1], _stream()),
          'mvae_embedding_swish_fwd')


def embedding_swish_bwd(idx, w, dact, dw, accumulate=False):
    _need_gpu(idx, w, dact, dw); _f32c(w, dact, dw)
    check(_lib.lib().mvae_embedding_swish_bwd(_ptr(idx), _index_kind(idx), _ptr(w), _ptr(dact), _ptr(dw),
                                              idx.numel(), w.shape[0], w.shape[1],
                                              ACCUMULATE if accumulate else 0, _stream()),
          'mvae_embedding_swish_bwd')


# ---------------------------------------------------------------------------- PoE / KL
def _experts(mus, lvs):
    ex = _lib.Experts()
    for i, (m, v) in enumerate(zip(mus, lvs)):
        ex.mu[i] = m.data_ptr()
        ex.logvar[i] = v.data_ptr()
    return ex


def _expert_ld(mus, lvs):
    if len(mus) > _lib.MAX_EXPERTS:
        raise RuntimeError('at most %d experts' % _lib.MAX_EXPERTS)
    ld = mus[0].stride(0) if mus else 0
    for t in list(mus) + list(lvs):
        _need_gpu(t)
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) != ld:
            raise RuntimeError('expert mu/logvar must be fp32 [B,D] views with unit column stride and '
                               'a common row stride')
    return ld


_ALL_EXPERTS = {}


def all_experts_mask(n_experts, device):
    """Device mask [1] with the low ``n_experts`` bits set -- the kernel reads it as uint32, the tensor is int32 with
    the same bits ((1 << 32) - 1 does not fit torch.int32: MVAE_MAX_EXPERTS = 32 experts is the all-ones word, -1)."""
    if not 1 <= n_experts <= _lib.MAX_EXPERTS:
        raise RuntimeError('1..%d experts, got %d' % (_lib.MAX_EXPERTS, n_experts))
    capturing = torch.cuda.is_current_stream_capturing()
    key = (n_experts, str(device))
    if not capturing and key in _ALL_EXPERTS:
        return _ALL_EXPERTS[key]          # read-only: one fill launch per (count, device), not one per model() call
    bits = (1 << n_experts) - 1
    m = torch.full((1,), bits - (1 << 32) if bits >= (1 << 31) else bits, dtype=torch.int32, device=device)
    m._mvae_mask_ok = m._version          # non-empty by construction: _check_no_prior_masks need not read it back
    if not capturing:                     # (a tensor filled under capture only has its contents after a replay)
        _ALL_EXPERTS[key] = m
    return m


def _check_no_prior_masks(masks_dev, variant, n_experts):
    """MVAE_POE_NO_PRIOR: a term whose mask selects no expert is an empty product (0 / 0 precision: NaN mu, inf
    logvar, no error from the launch -- the C side cannot see a device mask).  Host-built masks are checked here;
    masks that only exist on the device (a captured graph's tables) are the caller's to keep non-empty."""
    if not str(variant).endswith('-noprior') or masks_dev.numel() > 64 or torch.cuda.is_current_stream_capturing():
        return
    # the read-back below is a blocking device-to-host copy: pay it once per mask tensor CONTENT, never per launch
    # (every model() call on the drop-in surface paid it for a mask that is non-empty by construction)
    if getattr(masks_dev, '_mvae_mask_ok', None) == masks_dev._version:
        return
    live = (1 << n_experts) - 1
    for t, m in enumerate(masks_dev.tolist()):
        if (m & live) == 0:
            raise RuntimeError('PoE without the built-in prior: term %d selects no expert (mask %#x)' % (t, m & 0xffffffff))
    try:
        masks_dev._mvae_mask_ok = masks_dev._version
    except AttributeError:
        pass


def poe_fwd(mus, lvs, masks_dev, noise, mu, logvar, z, kl, variant):
    """mus/lvs: per-expert [B,D] (possibly column slices of a [B,2D] head); masks_dev int32/uint32 [T]."""
    ld = _expert_ld(mus, lvs)
    _need_gpu(masks_dev, noise, mu, logvar, z, kl); _f32c(noise, mu, logvar, z, kl)
    T, B, D = mu.shape
    ex = _experts(mus, lvs)
    _check_no_prior_masks(masks_dev, variant, len(mus))
    check(_lib.lib().mvae_poe_fwd(ctypes.byref(ex), ld, len(mus), _ptr(masks_dev), T, _ptr(noise), _ptr(mu),
                                  _ptr(logvar), _ptr(z), _ptr(kl), B, D, _lib.POE_VARIANT[variant],
                                  _stream()), 'mvae_poe_fwd')


def poe_fwd_draw(mus, lvs, masks_dev, noise_out, seed, counter_dev, counter_offset, mu, logvar, z, kl, variant):
    """``poe_fwd`` drawing its own eps: the values ``philox_fill(noise_out, seed, counter_dev, counter_offset)`` would
    have written, generated in the launch and stored to ``noise_out`` for the backward."""
    ld = _expert_ld(mus, lvs)
    _need_gpu(masks_dev, noise_out, counter_dev, mu, logvar, z, kl); _f32c(noise_out, mu, logvar, z, kl)
    T, B, D = mu.shape
    ex = _experts(mus, lvs)
    check(_lib.lib().mvae_poe_fwd_draw(ctypes.byref(ex), ld, len(mus), _ptr(masks_dev), T, _ptr(noise_out), seed,
                                       _ptr(counter_dev), int(counter_offset), _ptr(mu), _ptr(logvar), _ptr(z),
                                       _ptr(kl), B, D, _lib.POE_VARIANT[variant], _stream()), 'mvae_poe_fwd_draw')


def poe_bwd(mus, lvs, masks_dev, noise, mu, logvar, dz, dmu, dlogvar, dkl, g_mus, g_lvs, variant,
            dkl_per_term=False):
    """dkl: [T,B] row gradients of the KL output, or with dkl_per_term a [T] table (beta/B)."""
    ld = _expert_ld(mus, lvs)
    ldg = _expert_ld(g_mus, g_lvs)
    _need_gpu(masks_dev, noise, mu, logvar, dz, dmu, dlogvar, dkl)
    _f32c(noise, mu, logvar, dz, dmu, dlogvar, dkl)
    T, B, D = mu.shape
    ex = _experts(mus, lvs)
    gr = _lib.ExpertGrads()
    for i, (m, v) in enumerate(zip(g_mus, g_lvs)):
        gr.dmu[i] = m.data_ptr()
        gr.dlogvar[i] = v.data_ptr()
    check(_lib.lib().mvae_poe_bwd(ctypes.byref(ex), ld, len(mus), _ptr(masks_dev), T, _ptr(noise), _ptr(mu),
                                  _ptr(logvar), _ptr(dz), _ptr(dmu), _ptr(dlogvar), _ptr(dkl),
                                  1 if dkl_per_term else 0, ctypes.byref(gr), ldg, B, D, _lib.POE_VARIANT[variant], _stream()),
          'mvae_poe_bwd')


def poe_bwd_split(mus, lvs, masks_dev, noise, mu, logvar, dz_a, slots_a, dz_b, slots_b, dkl, g_mus, g_lvs, variant,
                  dkl_per_term=False):
    """poe_bwd with the latent gradient in two buffers: dz of term t = dz_a[slots_a[t]] + dz_b[slots_b[t]]
    (slot -1: not in that buffer); dz_* are [n_slots, B, D]."""
    ld = _expert_ld(mus, lvs)
    ldg = _expert_ld(g_mus, g_lvs)
    _need_gpu(masks_dev, noise, mu, logvar, dz_a, dz_b, dkl)
    _f32c(noise, mu, logvar, dz_a, dz_b, dkl)
    T, B, D = mu.shape
    if len(slots_a) != T or len(slots_b) != T:
        raise RuntimeError('one slot per term')
    for dz, slots in ((dz_a, slots_a), (dz_b, slots_b)):
        if dz.numel() != (max(slots) + 1) * B * D:
            raise RuntimeError('latent-gradient buffer does not match its slots')
    ex = _experts(mus, lvs)
    gr = _lib.ExpertGrads()
    for i, (m, v) in enumerate(zip(g_mus, g_lvs)):
        gr.dmu[i] = m.data_ptr()
        gr.dlogvar[i] = v.data_ptr()
    sa = (ctypes.c_int * T)(*[int(x) for x in slots_a])
    sb = (ctypes.c_int * T)(*[int(x) for x in slots_b])
    check(_lib.lib().mvae_poe_bwd_split(ctypes.byref(ex), ld, len(mus), _ptr(masks_dev), T, _ptr(noise), _ptr(mu),
                                        _ptr(logvar), _ptr(dz_a), sa, _ptr(dz_b), sb, _ptr(dkl),
                                        1 if dkl_per_term else 0, ctypes.byref(gr), ldg, B, D,
                                        _lib.POE_VARIANT[variant], _stream()), 'mvae_poe_bwd_split')


def kl_rows_fwd(mu, logvar, kl):
    _need_gpu(mu, logvar, kl); _f32c(mu, logvar, kl)
    check(_lib.lib().mvae_kl_rows_fwd(_ptr(mu), _ptr(logvar), _ptr(kl), mu.shape[0], mu.shape[1], _stream()),
          'mvae_kl_rows_fwd')


def kl_rows_bwd(mu, logvar, dkl, dmu, dlogvar):
    _need_gpu(mu, logvar, dkl, dmu, dlogvar); _f32c(mu, logvar, dkl, dmu, dlogvar)
    check(_lib.lib().mvae_kl_rows_bwd(_ptr(mu), _ptr(logvar), _ptr(dkl), _ptr(dmu), _ptr(dlogvar),
                                      mu.shape[0], mu.shape[1], _stream()), 'mvae_kl_rows_bwd')


# ---------------------------------------------------------------------------- losses
def bce_rowsum_fwd(logits, target, rowsum, colw=None, drow=None, dlogits=None, rows_per_group=None,
                   target_rows=None, target_div=1, target_strides=None):
    """logits [R,P]; target [target_rows,P] broadcast over row groups; optional fused gradient."""
    _need_gpu(logits, target, rowsum, colw, drow, dlogits); _f32c(logits, target, rowsum, colw, drow, dlogits)
    R, P = logits.shape
    t_rs, t_cs = target_strides if target_strides is not None else (P, 1)
    check(_lib.lib().mvae_bce_rowsum_fwd(_ptr(logits), _ptr(target), _ptr(colw), _ptr(rowsum), _ptr(drow),
                                         _ptr(dlogits), R, P, rows_per_group or R,
                                         target_rows or target.shape[0], target_div, t_rs, t_cs, _stream()),
          'mvae_bce_rowsum_fwd')


def bce_rowsum_bwd(logits, target, drow, dlogits, colw=None, rows_per_group=None, target_rows=None,
                   target_div=1, target_strides=None):
    _need_gpu(logits, target, drow, dlogits, colw); _f32c(logits, target, drow, dlogits, colw)
    R, P = logits.shape
    t_rs, t_cs = target_strides if target_strides is not None else (P, 1)
    check(_lib.lib().mvae_bce_rowsum_bwd(_ptr(logits), _ptr(target), _ptr(colw), _ptr(drow), _ptr(dlogits),
                                         R, P, rows_per_group or R, target_rows or target.shape[0],
                                         target_div, t_rs, t_cs, _stream()), 'mvae_bce_rowsum_bwd')


def ce_fwd(logits, label, row, drow=None, dlogits=None, rows_per_group=None, label_rows=None):
    _need_gpu(logits, label, row, drow, dlogits); _f32c(logits, row, drow, dlogits)
    if label.dtype != torch.int64 or not label.is_contiguous():
        raise RuntimeError('labels must be contiguous int64')
    R, K = logits.shape
    check(_lib.lib().mvae_ce_fwd(_ptr(logits), _ptr(label), _ptr(row), _ptr(drow), _ptr(dlogits), R, K,
                                 rows_per_group or R, label_rows or label.shape[0], _stream()),
          'mvae_ce_fwd')


def ce_bwd(logits, label, drow, dlogits, rows_per_group=None, label_rows=None):
    _need_gpu(logits, label, drow, dlogits); _f32c(logits, drow, dlogits)
    R, K = logits.shape
    check(_lib.lib().mvae_ce_bwd(_ptr(logits), _ptr(label), _ptr(drow), _ptr(dlogits), R, K,
                                 rows_per_group or R, label_rows or label.shape[0], _stream()),
          'mvae_ce_bwd')


def group_sums(rows, coef, out, total, G, rows_per_group, accumulate=False):
    """out[g] (+)= coef[g] * sum(rows[g*rpg:(g+1)*rpg]); total[0] (+)= their sum."""
    _need_gpu(rows, coef, out, total); _f32c(rows, coef, out, total)
    check(_lib.lib().mvae_group_sums(_ptr(rows), _ptr(coef), _ptr(out), _ptr(total), G, rows_per_group,
                                     ACCUMULATE if accumulate else 0, _stream()), 'mvae_group_sums')


def elbo_reduce(parts, elbo, T, zero=None, counter_dev=None, counter_inc=0):
    """parts: [(rows, coef or None, term_of or None, first_term, groups, rows_per_group)] -- see
    mvae_elbo_reduce in include/mvae_hip.h.  Overwrites elbo[0..T]."""
    if not 0 < len(parts) <= _lib.ELBO_MAX_PARTS:
        raise RuntimeError('1..%d ELBO parts per launch' % _lib.ELBO_MAX_PARTS)
    arr = (_lib.ElboPart * len(parts))()
    for q, (rows, coef, term_of, first, groups, rpg) in enumerate(parts):
        _need_gpu(rows, coef, term_of); _f32c(rows, coef)
        if rows.numel() < groups * rpg:
            raise RuntimeError('ELBO part %d: %d values for %d x %d' % (q, rows.numel(), groups, rpg))
        arr[q] = _lib.ElboPart(_ptr(rows), _ptr(coef), _ptr(term_of), int(first), int(groups), int(rpg))
    _need_gpu(elbo, zero, counter_dev); _f32c(elbo, zero)
    check(_lib.lib().mvae_elbo_reduce(arr, len(parts), _ptr(elbo), int(T), _ptr(zero),
                                      0 if zero is None else zero.numel(), _ptr(counter_dev), int(counter_inc),
                                      _stream()), 'mvae_elbo_reduce')


# ---------------------------------------------------------------------------- noise / optimiser
def philox_fill(out, seed, counter_dev, offset=0, keep_prob=None):
    """Standard normal (keep_prob None) or Bernoulli(keep_prob) draws at launch index *counter_dev + offset;
    the counter is NOT advanced (the fused step does that once per step in elbo_reduce)."""
    _need_gpu(out, counter_dev); _f32c(out)
    check(_lib.lib().mvae_philox_fill(_ptr(out), out.numel(), 0 if keep_prob is None else 1,
                                      0.0 if keep_prob is None else float(keep_prob), seed, _ptr(counter_dev),
                                      int(offset), _stream()), 'mvae_philox_fill')



def randn_(out, seed, counter_dev):
    _need_gpu(out, counter_dev); _f32c(out)
    check(_lib.lib().mvae_randn(_ptr(out), out.numel(), seed, _ptr(counter_dev), _stream()), 'mvae_randn')


def bernoulli_(out, keep_prob, seed, counter_dev):
    _need_gpu(out, counter_dev); _f32c(out)
    check(_lib.lib().mvae_bernoulli(_ptr(out), out.numel(), keep_prob, seed, _ptr(counter_dev), _stream()),
          'mvae_bernoulli')


def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              grad_scale=1.0):
    _need_gpu(param, grad, exp_avg, exp_avg_sq, step_dev); _f32c(param, grad, exp_avg, exp_avg_sq)
    check(_lib.lib().mvae_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                                    lr, beta1, beta2, eps, grad_scale, _ptr(step_dev), _stream()),
          'mvae_adam_step')


def adam_apply(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """Adam on (a range of) the arena at step *step_dev + 1, without advancing the counter."""
    _need_gpu(param, grad, exp_avg, exp_avg_sq, step_dev); _f32c(param, grad, exp_avg, exp_avg_sq)
    check(_lib.lib().mvae_adam_apply(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                                     lr, beta1, beta2, eps, grad_scale, _ptr(step_dev), _stream()),
          'mvae_adam_apply')


def adam_apply_at(param, grad, exp_avg, exp_avg_sq, step_dev, step_add, lr, beta1=0.9, beta2=0.999, eps=1e-8,
                  grad_scale=1.0):
    """Adam at step *step_dev + step_add, counter untouched (step_add 0: it was advanced earlier in the step)."""
    _need_gpu(param, grad, exp_avg, exp_avg_sq, step_dev); _f32c(param, grad, exp_avg, exp_avg_sq)
    check(_lib.lib().mvae_adam_apply_at(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                                        lr, beta1, beta2, eps, grad_scale, _ptr(step_dev), int(step_add), _stream()),
          'mvae_adam_apply_at')


def adam_apply_coef(param, grad, exp_avg, exp_avg_sq, coef2, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """``adam_apply_at(..., step_add=0)`` with the step's bias-correction factors read from ``coef2`` (``adam_prepare``)."""
    _need_gpu(param, grad, exp_avg, exp_avg_sq, coef2); _f32c(param, grad, exp_avg, exp_avg_sq, coef2)
    check(_lib.lib().mvae_adam_apply_coef(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                                          _ptr(coef2), beta1, beta2, eps, grad_scale, _stream()), 'mvae_adam_apply_coef')


def adam_prepare(step_dev, delta, lr, beta1, beta2, coef2):
    """*step_dev += delta; coef2[0:2] = Adam's two bias-correction factors at the new step (mvae_adam_prepare)."""
    _need_gpu(step_dev, coef2); _f32c(coef2)
    check(_lib.lib().mvae_adam_prepare(_ptr(step_dev), int(delta), lr, beta1, beta2, _ptr(coef2), _stream()),
          'mvae_adam_prepare')


def trace_marker(tag=0):
    """An empty kernel on the current stream (tools/step_by_shape.py splits a rocprofv3 kernel trace at these)."""
    check(_lib.lib().mvae_trace_marker(int(tag), _stream()), 'mvae_trace_marker')


def counter_add(counter_dev, delta):
    _need_gpu(counter_dev)
    check(_lib.lib().mvae_counter_add(_ptr(counter_dev), int(delta), _stream()), 'mvae_counter_add')


def fill_(out, value):
    _need_gpu(out); _f32c(out)
    check(_lib.lib().mvae_fill(_ptr(out), out.numel(), float(value), _stream()), 'mvae_fill')


def ingest(image_src, image_dst, label_src, label_dst, table_host, table_dst):
    """One launch: image and label batch to their static (graph-visible) buffers, the pinned host table block to
    its device block.  ``table_host`` is a pinned CPU tensor the kernel reads directly."""
    _need_gpu(image_src, image_dst, label_src, label_dst, table_dst)
    if not table_host.is_pinned():
        raise RuntimeError('ingest: the table source must be pinned host memory')
    for a, b in ((image_src, image_dst), (label_src, label_dst), (table_host, table_dst)):
        if a.dtype != b.dtype or a.numel() != b.numel() or not (a.is_contiguous() and b.is_contiguous()):
            raise RuntimeError('ingest: source and destination must be contiguous and alike')
    check(_lib.lib().mvae_ingest(_ptr(image_src), _ptr(image_dst), image_src.numel(),
                                 _ptr(label_src), _ptr(label_dst), label_src.numel() * label_src.element_size(),
                                 _ptr(table_host), _ptr(table_dst), table_host.numel() * table_host.element_size(),
                                 _stream()), 'mvae_ingest')


def ingest_ok(image_src, image_dst, label_src, label_dst):
    """Whether ``ingest`` takes this batch (else: plain copies)."""
    return (image_src.is_cuda and label_src.is_cuda and image_src.dtype == torch.float32 == image_dst.dtype
            and label_src.dtype == label_dst.dtype and image_src.is_contiguous() and label_src.is_contiguous()
            and image_src.numel() == image_dst.numel() and label_src.numel() == label_dst.numel()
            and image_src.numel() % 4 == 0 and (label_src.numel() * label_src.element_size()) % 4 == 0
            and image_src.data_ptr() % 16 == 0 and image_dst.data_ptr() % 16 == 0)


def dropout_fanout_fwd(h, masks, out, scale):
    """h [B,N], masks [G,B,N] -> out [G*B,N]"""
    _need_gpu(h, masks, out); _f32c(h, masks, out)
    G, B, N = masks.shape
    check(_lib.lib().mvae_dropout_fanout_fwd(_ptr(h), _ptr(masks), _ptr(out), scale, G, B, N, _stream()),
          'mvae_dropout_fanout_fwd')


def dropout_fanin_bwd(dout, masks, dh, scale):
    _need_gpu(dout, masks, dh); _f32c(dout, masks, dh)
    G, B, N = masks.shape
    check(_lib.lib().mvae_dropout_fanin_bwd(_ptr(dout), _ptr(masks), _ptr(dh), scale, G, B, N, _stream()),
          'mvae_dropout_fanin_bwd')


def bce_elem_fwd(logits, target, out):
    _need_gpu(logits, target, out); _f32c(logits, target, out)
    check(_lib.lib().mvae_bce_elem_fwd(_ptr(logits), _ptr(target), _ptr(out), logits.numel(), _stream()),
          'mvae_bce_elem_fwd')


def bce_elem_bwd(logits, target, g, dlogits, dtarget=None):
    _need_gpu(logits, target, g, dlogits, dtarget); _f32c(logits, target, g, dlogits, dtarget)
    check(_lib.lib().mvae_bce_elem_bwd(_ptr(logits), _ptr(target), _ptr(g), _ptr(dlogits), _ptr(dtarget),
                                       logits.numel(), _stream()), 'mvae_bce_elem_bwd')


# ---------------------------------------------------------------------------- GRU text stacks (MultiMNIST)
def _rows(t, what, width=None):
    """[rows, width] fp32 GPU view with unit column stride (a column range of a wider buffer is fine)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
        raise RuntimeError('%s must be a float32 [rows, width] GPU view with unit column stride' % what)
    if width is not None and t.shape[1] != width:
        raise RuntimeError('%s: width %d, expected %d' % (what, t.shape[1], width))
    return t


def gru_cell_fwd(gi, gh, h_prev, h_new, gates=None):
    B, H = _rows(h_prev, 'h_prev').shape
    _rows(gi, 'gi', 3 * H); _rows(gh, 'gh', 3 * H); _rows(h_new, 'h_new', H)
    if gates is not None:
        _need_gpu(gates); _f32c(gates)
    check(_lib.lib().mvae_gru_cell_fwd(_ptr(gi), gi.stride(0), _ptr(gh), gh.stride(0), _ptr(h_prev), h_prev.stride(0),
                                       _ptr(h_new), h_new.stride(0), _ptr(gates), B, H, _stream()), 'mvae_gru_cell_fwd')


def gru_cell_bwd(dh_new, dh_extra, gates, h_prev, dgi, dgh, dh_prev):
    B, H = _rows(h_prev, 'h_prev').shape
    _rows(dh_new, 'dh_new', H)
    if dh_extra is not None:
        _rows(dh_extra, 'dh_extra', H)
    _need_gpu(gates, dgi, dgh, dh_prev); _f32c(gates, dgi, dgh, dh_prev)
    check(_lib.lib().mvae_gru_cell_bwd(_ptr(dh_new), dh_new.stride(0), _ptr(dh_extra),
                                       0 if dh_extra is None else dh_extra.stride(0), _ptr(gates), _ptr(h_prev),
                                       h_prev.stride(0), _ptr(dgi), _ptr(dgh), _ptr(dh_prev), B, H, _stream()),
          'mvae_gru_cell_bwd')


def _idx64(idx):
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.dim() == 1):
        raise RuntimeError('embedding index must be a 1-D int64 GPU tensor (a column of text[B, L] is fine)')
    return max(int(idx.stride(0)), 1)


def embedding_fwd(idx, w, out, swish=False):
    _need_gpu(w); _f32c(w)
    R, width = _rows(out, 'out', w.shape[1]).shape
    check(_lib.lib().mvae_embedding_fwd(_ptr(idx), _idx64(idx), _ptr(w), _ptr(out), out.stride(0), idx.numel(),
                                        w.shape[0], width, ACT_SWISH if swish else 0, _stream()), 'mvae_embedding_fwd')


def embedding_bwd(idx, w, dout, dw, swish=False, accumulate=False):
    _need_gpu(w, dw); _f32c(w, dw)
    _rows(dout, 'dout', w.shape[1])
    flags = (ACT_SWISH if swish else 0) | (ACCUMULATE if accumulate else 0)
    check(_lib.lib().mvae_embedding_bwd(_ptr(idx), _idx64(idx), _ptr(w), _ptr(dout), dout.stride(0), _ptr(dw),
                                        idx.numel(), w.shape[0], w.shape[1], flags, _stream()), 'mvae_embedding_bwd')


def copy2d(src, dst, mask=None, scale=1.0, accumulate=False):
    rows, cols = _rows(src, 'src').shape
    _rows(dst, 'dst', cols)
    if mask is not None:
        _rows(mask, 'mask', cols)
    check(_lib.lib().mvae_copy2d(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), _ptr(mask),
                                 0 if mask is None else mask.stride(0), float(scale), rows, cols,
                                 ACCUMULATE if accumulate else 0, _stream()), 'mvae_copy2d')


def argmax_rows(x, out):
    R, K_ = _rows(x, 'x').shape
    if not (out.is_cuda and out.dtype == torch.int64 and out.is_contiguous() and out.numel() == R):
        raise RuntimeError('argmax_rows wants a contiguous int64 GPU output of one entry per row')
    check(_lib.lib().mvae_argmax_rows(_ptr(x), x.stride(0), _ptr(out), R, K_, _stream()), 'mvae_argmax_rows')

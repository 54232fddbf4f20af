"""GPU: every HIP kernel against a plain torch-CPU fp32 reference of the same op, called
through the C ABI (mvae_amd.kernels -> ctypes -> libmvae_hip.so).  Tolerance: 1e-4 relative
(max |err| / max |ref|), the north_star bound; most ops land near 1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mvae_amd
from mvae_amd import kernels as K
from oracle import functional as OF
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def g(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def dev(t):
    return None if t is None else t.to(DEV)


def swish(x):
    return x * torch.sigmoid(x)


def swish_grad(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


# ----------------------------------------------------------------------------- Linear
LIN_SHAPES = [(128, 512, 784), (512, 128, 512), (37, 10, 512), (256, 512, 18), (300, 1, 512),
              (64, 200, 6400), (1024, 6272, 512), (130, 100, 200), (3, 5, 7),
              # the MNIST step's shapes at batch 512 (rows of two ELBO terms = 1024): these run on the small
              # 64x32 / 32x64 / 32x32 layouts with in-block k-groups instead of a split reduction
              (1024, 512, 512), (512, 512, 512), (512, 512, 784), (1024, 784, 512), (1024, 512, 64),
              (1024, 64, 512), (1000, 500, 516), (72, 36, 100)]


@pytest.mark.parametrize('M,N,Kd', LIN_SHAPES)
def test_linear_fwd(M, N, Kd):
    x, w, b = g(M, Kd, seed=1), g(N, Kd, seed=2, scale=Kd ** -0.5), g(N, seed=3)
    pre = torch.empty(M, N, device=DEV); act = torch.empty(M, N, device=DEV)
    K.linear_fwd(dev(x), dev(w), dev(b), pre, act)
    ref = x @ w.t() + b
    assert_close(pre, ref, 'linear pre')
    assert_close(act, swish(ref), 'linear act')
    # dropout mask fused on the activation, no pre buffer
    mask = (torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.9).float()
    act2 = torch.empty(M, N, device=DEV)
    K.linear_fwd(dev(x), dev(w), dev(b), None, act2, dev(mask), 1 / 0.9)
    assert_close(act2, swish(ref) * (mask / 0.9), 'linear act*mask')
    # no bias
    K.linear_fwd(dev(x), dev(w), None, pre, None)
    assert_close(pre, x @ w.t(), 'linear nobias')


@pytest.mark.parametrize('M,N,Kd', LIN_SHAPES)
def test_linear_dgrad(M, N, Kd):
    dy, w = g(M, N, seed=5), g(N, Kd, seed=6, scale=N ** -0.5)
    pre_in = g(M, Kd, seed=7)
    mask = (torch.rand(M, Kd, generator=torch.Generator().manual_seed(8)) < 0.9).float()
    dx = torch.empty(M, Kd, device=DEV)
    K.linear_dgrad(dev(dy), dev(w), dx)
    ref = dy @ w
    assert_close(dx, ref, 'dgrad plain')
    K.linear_dgrad(dev(dy), dev(w), dx, dev(pre_in))
    assert_close(dx, ref * swish_grad(pre_in), 'dgrad * swish\'')
    K.linear_dgrad(dev(dy), dev(w), dx, dev(pre_in), dev(mask), 1 / 0.9)
    assert_close(dx, ref * (mask / 0.9) * swish_grad(pre_in), 'dgrad * mask * swish\'')
    base = g(M, Kd, seed=9)
    dx2 = dev(base).clone()
    K.linear_dgrad(dev(dy), dev(w), dx2, accumulate=True)
    assert_close(dx2, base + ref, 'dgrad accumulate')


@pytest.mark.parametrize('M,N,Kd', LIN_SHAPES)
def test_linear_wgrad(M, N, Kd):
    dy, x = g(M, N, seed=10), g(M, Kd, seed=11)
    dw = torch.empty(N, Kd, device=DEV); db = torch.empty(N, device=DEV)
    K.linear_wgrad(dev(dy), dev(x), dw, db)
    assert_close(dw, dy.t() @ x, 'wgrad dw')
    assert_close(db, dy.sum(0), 'wgrad db')
    K.linear_wgrad(dev(dy), dev(x), dw, db, accumulate=True)
    assert_close(dw, 2 * (dy.t() @ x), 'wgrad dw accumulate')
    assert_close(db, 2 * dy.sum(0), 'wgrad db accumulate')
    dw2 = torch.empty(N, Kd, device=DEV)
    K.linear_wgrad(dev(dy), dev(x), dw2, None)
    assert_close(dw2, dy.t() @ x, 'wgrad no bias')


@pytest.mark.parametrize('shapes', [
    [(1024, 512, 64), (1024, 512, 512), (1024, 512, 512), (1024, 784, 512)],       # MNIST image decoder
    [(512, 128, 512), (512, 512, 512), (512, 512, 784)],                            # MNIST image encoder
    [(100, 33, 70), (7, 40, 40), (257, 96, 32)],                                    # ragged everything
    [(64, 32, 32)] * 17,                                                            # more than one launch's worth
])
def test_linear_wgrad_batched(shapes):
    """Several Linear weight gradients in one launch == each one alone (same arithmetic: bit-equal)."""
    items, refs = [], []
    for q, (M, N, Kd) in enumerate(shapes):
        dy, x = g(M, N, seed=100 + q), g(M, Kd, seed=200 + q)
        dw = torch.full((N, Kd), 7.0, device=DEV)
        db = torch.full((N,), 3.0, device=DEV) if q % 3 != 2 else None
        acc = (q % 2 == 1)
        items.append((dev(dy), dev(x), dw, db, acc))
        refs.append((dy, x, acc))
    assert all(K.wgrad_batchable(it[0], it[1]) for it in items)
    K.linear_wgrad_batched(items)
    for (dyd, xd, dw, db, acc), (dy, x, _) in zip(items, refs):
        rw = dy.t() @ x + (7.0 if acc else 0.0)
        assert_close(dw, rw, 'batched dw')
        if db is not None:
            assert_close(db, dy.sum(0) + (3.0 if acc else 0.0), 'batched db')
        # the single-problem launch of the same problem gives the same bits (when it takes the direct path)
        dw1 = torch.full_like(dw, 7.0); db1 = None if db is None else torch.full_like(db, 3.0)
        K.linear_wgrad(dyd, xd, dw1, db1, accumulate=acc)
        assert_close(dw, dw1.cpu(), 'batched vs single', tol=1e-6)


def test_conv_repack_batched_equals_in_launch_repack():
    """Repacked weight copies made ahead in one launch (w = NULL to the compute launch) == the copies the launches
    make themselves; shapes with their own kernels report 0 floats and refuse w = NULL."""
    B = 3
    conv_w = dev(g(64, 32, 4, 4, seed=300, scale=0.05))          # Conv2d(32, 64), 32x32 input, stride 2
    convT_w = dev(g(128, 64, 4, 4, seed=301, scale=0.05))        # ConvTranspose2d(128, 64), 8x8 input
    n1 = K.conv_repack_floats(False, conv_w, B, 32, 32, 32, 64, 2, 1)
    n2 = K.conv_repack_floats(True, convT_w, B, 128, 8, 8, 64, 2, 1)
    assert n1 == 64 * 32 * 16 and n2 == 128 * 64 * 16
    wr1, wr2 = torch.empty(n1, device=DEV), torch.empty(n2, device=DEV)
    K.conv_repack_batched([(conv_w, wr1, False, 32, 64, 2, 1), (convT_w, wr2, True, 128, 64, 2, 1)])
    dy = dev(g(B, 64, 16, 16, seed=302)); dx_a = torch.empty(B, 32, 32, 32, device=DEV); dx_b = torch.empty_like(dx_a)
    K.conv2d_dgrad(dy, conv_w, dx_a, None, 2, 1)
    K.conv2d_dgrad(dy, conv_w, dx_b, None, 2, 1, wr=wr1)
    assert torch.equal(dx_a, dx_b)
    x = dev(g(B, 128, 8, 8, seed=303)); y_a = torch.empty(B, 64, 16, 16, device=DEV); y_b = torch.empty_like(y_a)
    K.convT2d_fwd(x, convT_w, y_a, None, 2, 1)
    K.convT2d_fwd(x, convT_w, y_b, None, 2, 1, wr=wr2)
    assert torch.equal(y_a, y_b)
    # direct-kernel shapes: the 5x5 stride-1 transposed conv and the 3-channel output
    s1_w = dev(g(256, 128, 4, 4, seed=304, scale=0.05))
    assert K.conv_repack_floats(True, s1_w, B, 256, 5, 5, 128, 1, 0) == 0
    small_w = dev(g(32, 3, 4, 4, seed=305, scale=0.05))
    assert K.conv_repack_floats(True, small_w, B, 32, 32, 32, 3, 2, 1) == 0
    with pytest.raises(RuntimeError):
        K.convT2d_fwd(dev(g(B, 32, 32, 32, seed=306)), small_w, torch.empty(B, 3, 64, 64, device=DEV), None, 2, 1,
                      wr=torch.empty(32 * 3 * 16, device=DEV))


@pytest.mark.parametrize('shapes', [
    [(1024, 512, 64), (1024, 512, 512), (1024, 784, 512)],
    [(100, 33, 70), (7, 40, 40), (257, 96, 32)],
    [(64, 32, 32)] * 17,
])
def test_linear_wgrad_batched_adam_is_wgrad_then_adam(shapes):
    """The launch that updates its own outputs == a batch followed by Adam over the same arena ranges: parameters and
    both moments bit for bit GIVEN the gradients it produced -- incl. an update-only item for a gradient written earlier --
    and its gradients equal to the plain batch launch's to fp32 round-off (the plain launch has run on a different tile code
    since round 5 -- 64-wide wave tiles, another summation tree -- so the two are no longer the same bits)."""
    from mvae_amd import _lib
    sizes = []
    for (M, N, Kd) in shapes:
        sizes += [N * Kd, N]
    extra = 5120 + 3                      # the finished gradient (an Embedding's), ragged against the 1024-element tile
    offs, off = [], 0
    for n in sizes + [extra]:
        offs.append(off); off += (n + 3) // 4 * 4
    total = off

    def arenas(seed):
        return (dev(g(total, seed=seed, scale=0.1)), torch.zeros(total, device=DEV),
                dev(g(total, seed=seed + 1, scale=0.01)), dev(g(total, seed=seed + 2, scale=0.01)).abs())
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    results = []
    for fused in (True, False):
        param, grad, m, v = arenas(900)
        step = torch.full((1,), 6, dtype=torch.int64, device=DEV)
        coef = torch.zeros(2, device=DEV)
        o_extra = offs[-1]
        grad[o_extra:o_extra + extra] = dev(g(extra, seed=77))
        items = []
        for q, (M, N, Kd) in enumerate(shapes):
            dy, x = dev(g(M, N, seed=100 + q)), dev(g(M, Kd, seed=200 + q))
            ow, ob = offs[2 * q], offs[2 * q + 1]
            dw = grad[ow:ow + N * Kd].view(N, Kd)
            db = grad[ob:ob + N] if q % 3 != 2 else None
            items.append((dy, x, dw, db, False))
        K.adam_prepare(step, 1, lr, b1, b2, coef)
        assert int(step.item()) == 7
        if fused:
            st = _lib.AdamFuse(grad.data_ptr(), param.data_ptr(), m.data_ptr(), v.data_ptr(), coef.data_ptr(), b1, b2,
                               eps, 1.0)
            K.linear_wgrad_batched(items + [(None, None, grad[o_extra:o_extra + extra], None, False)], adam=st)
        else:
            K.linear_wgrad_batched(items)
            torch.cuda.synchronize()
            plain = grad.clone()
            grad.copy_(results[0][1])           # Adam below runs on exactly the gradients the fused launch produced
            for (dy, x, dw, db, _) in items:
                for t in (dw, db):
                    if t is not None:
                        lo = (t.data_ptr() - grad.data_ptr()) // 4
                        hi = lo + t.numel()
                        K.adam_apply_at(param[lo:hi], grad[lo:hi], m[lo:hi], v[lo:hi], step, 0, lr, b1, b2, eps)
            lo, hi = o_extra, o_extra + extra
            K.adam_apply_at(param[lo:hi], grad[lo:hi], m[lo:hi], v[lo:hi], step, 0, lr, b1, b2, eps)
        torch.cuda.synchronize()
        results.append((param.clone(), grad.clone(), m.clone(), v.clone()))
    for a, b, what in zip(results[0], results[1], ('param', 'grad', 'exp_avg', 'exp_avg_sq')):
        assert torch.equal(a, b), what
    # the plain batch launch against the fused one: the same sums in another order
    for q, (M, N, Kd) in enumerate(shapes):
        for o, n in ((offs[2 * q], N * Kd),) + (((offs[2 * q + 1], N),) if q % 3 != 2 else ()):
            a, b = plain[o:o + n], results[0][1][o:o + n]
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-30), 'gradient of item %d' % q
    # and the bias gradients nobody asked for (db = None) left their parameters alone
    p0 = arenas(900)[0]
    for q in range(len(shapes)):
        if q % 3 == 2:
            ob, N = offs[2 * q + 1], shapes[q][1]
            assert torch.equal(results[0][0][ob:ob + N], p0[ob:ob + N])
    # Adam against torch's own arithmetic on one weight (tolerance: its lerp / addcdiv round differently)
    M, N, Kd = shapes[0]
    w0, m0, v0 = [t[offs[0]:offs[0] + N * Kd].cpu().double() for t in arenas(900)[0:1] + arenas(900)[2:4]]
    gw = (g(M, N, seed=100).double().t() @ g(M, Kd, seed=200).double()).reshape(-1)
    m1 = b1 * m0 + (1 - b1) * gw
    v1 = b2 * v0 + (1 - b2) * gw * gw
    ref = w0 - (lr / (1 - b1 ** 7)) * m1 / (v1.sqrt() / (1 - b2 ** 7) ** 0.5 + eps)
    assert_close(results[0][0][offs[0]:offs[0] + N * Kd], ref.float(), 'fused Adam vs fp64')


def test_linear_wgrad_batched_adam_refuses_accumulation():
    from mvae_amd import _lib
    dy, x = dev(g(64, 32, seed=1)), dev(g(64, 32, seed=2))
    buf = torch.zeros(4, 32 * 32, device=DEV)
    coef = torch.zeros(2, device=DEV)
    st = _lib.AdamFuse(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr(), coef.data_ptr(),
                       0.9, 0.999, 1e-8, 1.0)
    with pytest.raises(RuntimeError):
        K.linear_wgrad_batched([(dy, x, buf[0].view(32, 32), None, True)], adam=st)
    with pytest.raises(RuntimeError):
        K.linear_wgrad_batched([(None, None, buf[0], None, False)])          # update-only items need adam=


def test_linear_wgrad_batched_rejects_shared_gradient():
    dy, x = dev(g(64, 32, seed=1)), dev(g(64, 32, seed=2))
    dw = torch.empty(32, 32, device=DEV)
    with pytest.raises(RuntimeError):
        K.linear_wgrad_batched([(dy, x, dw, None, False), (dy, x, dw, None, True)])


def test_linear_strided_views():
    """Row-strided operands: column slices of a [B, 2D] head and a column of the [rows, 18] logits."""
    M, Kd = 96, 64
    big = g(M, 2 * Kd, seed=12)
    w, b = g(32, Kd, seed=13, scale=0.1), g(32, seed=14)
    xs = dev(big)[:, Kd:]
    pre = torch.empty(M, 32, device=DEV)
    K.linear_fwd(xs, dev(w), dev(b), pre, None)
    assert_close(pre, big[:, Kd:] @ w.t() + b, 'strided x')
    out = torch.zeros(M, 18, device=DEV)
    w1, b1 = g(1, Kd, seed=15), g(1, seed=16)
    K.linear_fwd(dev(big)[:, :Kd], dev(w1), dev(b1), out[:, 5:6], None)
    ref = torch.zeros(M, 18); ref[:, 5:6] = big[:, :Kd] @ w1.t() + b1
    assert_close(out, ref, 'strided y (ldy=18)')


# ----------------------------------------------------------------------------- Conv / ConvT
CONV_CASES = [(4, 3, 64, 32, 2, 1), (3, 32, 32, 64, 2, 1), (2, 64, 16, 128, 2, 1), (2, 128, 8, 256, 1, 0),
              (5, 1, 28, 64, 2, 1), (2, 64, 14, 128, 2, 1), (1, 5, 6, 7, 1, 0), (3, 2, 4, 3, 2, 1),
              (130, 3, 64, 32, 2, 1), (37, 1, 28, 64, 2, 1)]      # more (image, row) units than waves of the small-Cin wgrad


# ----------------------------------------------------------------------------- grouped Linear / Embedding
GROUPED_SHAPES = [(18, 256, 512, 512), (18, 768, 1, 512), (18, 768, 512, 100), (18, 64, 200, 512),
                  (3, 37, 10, 12), (5, 130, 7, 33)]


def _arena_like(G, shape, pad, seed):
    """G tensors of ``shape`` at a uniform stride inside one flat buffer (like the experts' slices
    of the parameter arena); returns (flat device buffer, [views], stride in floats)."""
    n = int(np.prod(shape))
    stride = (n + pad + 3) // 4 * 4
    flat = (g(G * stride, seed=seed) * 0.1).to(DEV)
    views = [flat[i * stride:i * stride + n].view(*shape) for i in range(G)]
    return flat, views, stride


@pytest.mark.parametrize('G,M,N,Kd', GROUPED_SHAPES)
def test_linear_grouped(G, M, N, Kd):
    x = g(G, M, Kd, seed=1)
    _, ws, w_gs = _arena_like(G, (N, Kd), 40, 2)
    _, bs, b_gs = _arena_like(G, (N,), 8, 3)
    w = torch.stack([t.cpu() for t in ws]); b = torch.stack([t.cpu() for t in bs])
    pre_ref = torch.einsum('gmk,gnk->gmn', x, w) + b[:, None, :]
    pre = torch.empty(G, M, N, device=DEV); act = torch.empty_like(pre)
    K.linear_fwd_grouped(dev(x), ws[0], w_gs, bs[0], b_gs, pre, act)
    assert_close(pre, pre_ref, 'grouped fwd pre')
    assert_close(act, swish(pre_ref), 'grouped fwd act')
    # dgrad with the producer's swish' fused
    dy = g(G, M, N, seed=4); pre_in = g(G, M, Kd, seed=5)
    dx = torch.empty(G, M, Kd, device=DEV)
    K.linear_dgrad_grouped(dev(dy), ws[0], w_gs, dx, dev(pre_in))
    assert_close(dx, torch.einsum('gmn,gnk->gmk', dy, w) * swish_grad(pre_in), 'grouped dgrad')
    # wgrad + bias gradient, overwrite then accumulate
    dwf, dws, dw_gs = _arena_like(G, (N, Kd), 40, 6)
    dbf, dbs, db_gs = _arena_like(G, (N,), 8, 7)
    dwf_before = dwf.clone()
    K.linear_wgrad_grouped(dev(dy), dev(x), dws[0], dw_gs, dbs[0], db_gs)
    dw_ref = torch.einsum('gmn,gmk->gnk', dy, x); db_ref = dy.sum(1)
    assert_close(torch.stack([t for t in dws]), dw_ref, 'grouped wgrad')
    assert_close(torch.stack([t for t in dbs]), db_ref, 'grouped bias grad')
    n = N * Kd
    for i in range(G):      # the padding between the experts' slices is untouched
        assert torch.equal(dwf[i * dw_gs + n:(i + 1) * dw_gs], dwf_before[i * dw_gs + n:(i + 1) * dw_gs])
    K.linear_wgrad_grouped(dev(dy), dev(x), dws[0], dw_gs, dbs[0], db_gs, accumulate=True)
    assert_close(torch.stack([t for t in dws]), 2 * dw_ref, 'grouped wgrad accumulate')
    assert_close(torch.stack([t for t in dbs]), 2 * db_ref, 'grouped bias grad accumulate')


@pytest.mark.parametrize('M,N,Kd', [(1024, 512, 512), (1024, 512, 64), (96, 512, 512), (37, 10, 24)])
def test_linear_pair(M, N, Kd):
    """Two same-shaped problems at unrelated addresses (second one allocated FIRST: negative pointer
    difference) as one grouped launch, including the shapes whose reduction gets split."""
    second = [dev(g(M, Kd, seed=2)), dev(g(N, Kd, seed=4) * 0.1), dev(g(N, seed=6))]
    first = [dev(g(M, Kd, seed=1)), dev(g(N, Kd, seed=3) * 0.1), dev(g(N, seed=5))]
    x, w, b = (first[0], second[0]), (first[1], second[1]), (first[2], second[2])
    pre = (torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV))
    act = (torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV))
    if (pre[1].data_ptr() - pre[0].data_ptr()) != (act[1].data_ptr() - act[0].data_ptr()):
        buf = torch.empty(4, M, N, device=DEV)          # same spacing for pre and act
        pre, act = (buf[0], buf[2]), (buf[1], buf[3])
    K.linear_fwd_pair(x, w, b, pre, act)
    for i in range(2):
        ref = x[i].cpu() @ w[i].cpu().t() + b[i].cpu()
        assert_close(pre[i], ref, 'pair fwd pre %d' % i)
        assert_close(act[i], swish(ref), 'pair fwd act %d' % i)
    dy = (dev(g(M, N, seed=7)), dev(g(M, N, seed=8)))
    pin = torch.stack([g(M, Kd, seed=9), g(M, Kd, seed=10)]).to(DEV)
    dx = torch.empty(2, M, Kd, device=DEV)
    K.linear_dgrad_pair(dy, w, (dx[0], dx[1]), (pin[0], pin[1]))
    for i in range(2):
        assert_close(dx[i], (dy[i].cpu() @ w[i].cpu()) * swish_grad(pin[i].cpu()), 'pair dgrad %d' % i)
    dw = (torch.empty(N, Kd, device=DEV), torch.empty(N, Kd, device=DEV))
    db = (torch.empty(N, device=DEV), torch.empty(N, device=DEV))
    if (dw[1].data_ptr() - dw[0].data_ptr()) % 16:
        pytest.skip('allocator gave a 4-byte-granular spacing')
    K.linear_wgrad_pair(dy, x, dw, db)
    K.linear_wgrad_pair(dy, x, dw, db, accumulate=True)
    for i in range(2):
        assert_close(dw[i], 2 * (dy[i].cpu().t() @ x[i].cpu()), 'pair wgrad %d' % i)
        assert_close(db[i], 2 * dy[i].cpu().sum(0), 'pair bias grad %d' % i)


def test_embedding_grouped():
    G, R, W = 18, 300, 512
    idx = torch.randint(0, 2, (R, G), generator=torch.Generator().manual_seed(1)).float()
    _, ws, w_gs = _arena_like(G, (2, W), 12, 2)
    w = torch.stack([t.cpu() for t in ws])
    act = torch.empty(G, R, W, device=DEV)
    K.embedding_swish_fwd_grouped(dev(idx), ws[0], w_gs, act)
    ref = torch.stack([swish(w[i][idx[:, i].long()]) for i in range(G)])
    assert_close(act, ref, 'grouped embedding fwd')
    dact = g(G, R, W, seed=3)
    dwf, dws, dw_gs = _arena_like(G, (2, W), 12, 4)
    assert dw_gs == w_gs
    K.embedding_swish_bwd_grouped(dev(idx), ws[0], w_gs, dev(dact), dws[0])
    dref = []
    for i in range(G):
        d = torch.zeros(2, W)
        d.index_add_(0, idx[:, i].long(), dact[i])
        dref.append(d * swish_grad(w[i]))
    assert_close(torch.stack([t for t in dws]), torch.stack(dref), 'grouped embedding bwd')



@pytest.mark.parametrize('B,Cin,H,Cout,s,p', CONV_CASES)
def test_conv2d(B, Cin, H, Cout, s, p):
    x = g(B, Cin, H, H, seed=20).requires_grad_()
    w = g(Cout, Cin, 4, 4, seed=21, scale=(Cin * 16) ** -0.5).requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    dy = g(*y.shape, seed=22)
    y.backward(dy)
    pre = torch.empty(*y.shape, device=DEV); act = torch.empty(*y.shape, device=DEV)
    K.conv2d_fwd(dev(x.detach()), dev(w.detach()), pre, act, s, p)
    assert_close(pre, y, 'conv fwd')
    assert_close(act, swish(y.detach()), 'conv fwd act')
    dx = torch.empty(*x.shape, device=DEV)
    K.conv2d_dgrad(dev(dy), dev(w.detach()), dx, None, s, p)
    assert_close(dx, x.grad, 'conv dgrad')
    pre_in = g(*x.shape, seed=23)
    K.conv2d_dgrad(dev(dy), dev(w.detach()), dx, dev(pre_in), s, p)
    assert_close(dx, x.grad * swish_grad(pre_in), 'conv dgrad * swish\'')
    dw = torch.empty(*w.shape, device=DEV)
    K.conv2d_wgrad(dev(dy), dev(x.detach()), dw, s, p)
    assert_close(dw, w.grad, 'conv wgrad')
    K.conv2d_wgrad(dev(dy), dev(x.detach()), dw, s, p, accumulate=True)
    assert_close(dw, 2 * w.grad, 'conv wgrad accumulate')


@pytest.mark.parametrize('B,Cin,H,Cout', [(2049, 6, 14, 1), (2049, 8, 14, 1), (2047, 64, 14, 1), (512, 8, 32, 3), (1024, 4, 28, 2)])
def test_small_channel_transposed_conv_through_lds(B, Cin, H, Cout):
    """<= 4 output channels and >= 1024 blocks: the launch that stages the input rows through LDS (conv.hip:
    convT_small3_kernel) -- whole images per block with a ragged last block, 8 / 2 / 4 channels per trip; row bands of
    one image, the last band partial -- as a transposed conv forward and as the data gradient of the mirrored conv with
    the producer's Swish' folded in.  Channel counts that are multiples of 4 take the LDS-DMA form (convT_small3d_kernel:
    two / three pieces per wave, borders and the missing image of a ragged last block as out-of-range pieces), 6 the
    register-staged one."""
    x = g(B, Cin, H, H, seed=40)
    w = g(Cin, Cout, 4, 4, seed=41, scale=(Cin * 4) ** -0.5)
    y = F.conv_transpose2d(x, w, None, 2, 1)
    pre = torch.empty(*y.shape, device=DEV); act = torch.empty(*y.shape, device=DEV)
    K.convT2d_fwd(dev(x), dev(w), pre, act, 2, 1)
    assert_close(pre, y, 'convT fwd through LDS')
    assert_close(act, swish(y), 'convT fwd through LDS, act')
    # Conv2d(Cout, Cin): its data gradient is the same launch on dy = x
    pre_in = g(*y.shape, seed=42)
    dx = torch.empty(*y.shape, device=DEV)
    K.conv2d_dgrad(dev(x), dev(w), dx, dev(pre_in), 2, 1)
    assert_close(dx, y * swish_grad(pre_in), "conv dgrad * swish' through LDS")


CONVT_CASES = [(3, 256, 5, 128, 1, 0), (2, 128, 8, 64, 2, 1), (2, 64, 16, 32, 2, 1), (2, 32, 32, 3, 2, 1),
               (4, 128, 7, 64, 2, 1), (3, 64, 14, 1, 2, 1), (2, 3, 2, 5, 1, 0),
               (1923, 256, 5, 128, 1, 0)]     # >= 6144 blocks: the wide form of convT_s1_kernel, ragged last image group


@pytest.mark.parametrize('B,Cin,H,Cout,s,p', CONVT_CASES)
def test_conv_transpose2d(B, Cin, H, Cout, s, p):
    x = g(B, Cin, H, H, seed=30).requires_grad_()
    w = g(Cin, Cout, 4, 4, seed=31, scale=(Cin * 4) ** -0.5).requires_grad_()
    y = F.conv_transpose2d(x, w, None, s, p)
    dy = g(*y.shape, seed=32)
    y.backward(dy)
    pre = torch.empty(*y.shape, device=DEV); act = torch.empty(*y.shape, device=DEV)
    K.convT2d_fwd(dev(x.detach()), dev(w.detach()), pre, act, s, p)
    assert_close(pre, y, 'convT fwd')
    assert_close(act, swish(y.detach()), 'convT fwd act')
    dx = torch.empty(*x.shape, device=DEV)
    K.convT2d_dgrad(dev(dy), dev(w.detach()), dx, None, s, p)
    assert_close(dx, x.grad, 'convT dgrad')
    dw = torch.empty(*w.shape, device=DEV)
    K.convT2d_wgrad(dev(dy), dev(x.detach()), dw, s, p)
    assert_close(dw, w.grad, 'convT wgrad')


# ----------------------------------------------------------------------------- BatchNorm
# the last three rows of the first line and all of the second: the single-launch forms of norm.hip (slices of <= 16 K
# elements: the float4, the odd-width and the BatchNorm1d kernel) at CelebA's real layer shapes, their group limits
# (forward 3 / backward 2 groups, then the two-launch path), a ragged column block and > 64 rows per thread class
@pytest.mark.parametrize('G,B,C,spatial', [(1, 16, 64, (16, 16)), (3, 8, 32, (32, 32)), (2, 6, 256, (5, 5)),
                                           (3, 32, 512, ()), (1, 256, 512, ()), (2, 64, 128, (8, 8)),
                                           (3, 256, 128, (8, 8)), (2, 256, 128, (8, 8)), (1, 256, 256, (5, 5)),
                                           (1, 256, 64, (16, 16)), (4, 8, 32, (8, 8)), (2, 300, 40, ()),
                                           (3, 256, 512, ()), (4, 16, 48, ()), (3, 40, 24, (5, 5)), (1, 3, 8, (2, 2)),
                                           # more groups / larger slices than the all-groups form takes (two-launch path), and
                                           # the per-slice forward form (norm.hip kind 3: 16 K .. 64 K elements, >= 512 slices)
                                           (1, 256, 64, (16, 16)), (2, 256, 16, (16, 16)), (5, 64, 32, (8, 8)),
                                           (18, 32, 16, (16, 16)), (3, 50, 8, (32, 32)), (2, 70, 8, (32, 32)),
                                           (3, 4, 128, (8, 8)), (18, 4, 64, (16, 16)), (4, 2, 16, (4, 4)),
                                           (32, 68, 16, (16, 16)), (18, 128, 32, (16, 16)), (16, 20, 32, (32, 32))])
@pytest.mark.parametrize('act', [True, False])
def test_batchnorm_train(G, B, C, spatial, act):
    x = g(G * B, C, *spatial, seed=40) * 1.7 + 0.4
    gamma, beta = 1 + 0.1 * g(C, seed=41), 0.1 * g(C, seed=42)
    dy = g(*x.shape, seed=43)
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(); gr = gamma.clone().requires_grad_(); br = beta.clone().requires_grad_()
    outs = []
    for gi in range(G):
        for _ in range(2):   # n_updates = 2: the second call must not change the output
            o = F.batch_norm(xr[gi * B:(gi + 1) * B], rm, rv, gr, br, True, 0.1, 1e-5)
        outs.append(swish(o) if act else o)
    yref = torch.cat(outs)
    yref.backward(dy)
    y = torch.empty(*x.shape, device=DEV)
    sm = torch.empty(G, C, device=DEV); si = torch.empty(G, C, device=DEV)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    K.bn_train_fwd(dev(x), dev(gamma), dev(beta), y, sm, si, rmd, rvd, G, n_updates=2, swish=act)
    assert_close(y, yref, 'bn fwd')
    assert_close(rmd, rm, 'bn running_mean', tol=1e-5)
    assert_close(rvd, rv, 'bn running_var', tol=1e-5)
    dx = torch.empty(*x.shape, device=DEV)
    dg = torch.empty(C, device=DEV); db = torch.empty(C, device=DEV)
    K.bn_train_bwd(dev(dy), dev(x), dev(gamma), dev(beta), sm, si, dx, dg, db, G, swish=act)
    assert_close(dx, xr.grad, 'bn dx')
    assert_close(dg, gr.grad, 'bn dgamma')
    assert_close(db, br.grad, 'bn dbeta')
    K.bn_train_bwd(dev(dy), dev(x), dev(gamma), dev(beta), sm, si, dx, dg, db, G, swish=act, accumulate=True)
    assert_close(dg, 2 * gr.grad, 'bn dgamma accumulate')
    # statistics-only call (the decoder passes that exist for their running-statistics side effect): no output, same
    # saved and running statistics
    sm2 = torch.empty(G, C, device=DEV); si2 = torch.empty(G, C, device=DEV)
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    K.bn_train_fwd(dev(x), dev(gamma), dev(beta), None, sm2, si2, rm2, rv2, G, n_updates=2, swish=act)
    assert torch.equal(sm2, sm) and torch.equal(si2, si) and torch.equal(rm2, rmd) and torch.equal(rv2, rvd)


def test_batchnorm_eval():
    x = g(6, 32, 8, 8, seed=44)
    gamma, beta = 1 + 0.1 * g(32, seed=45), 0.1 * g(32, seed=46)
    rm, rv = 0.3 * g(32, seed=47), 1 + 0.2 * torch.rand(32, generator=torch.Generator().manual_seed(48))
    y = torch.empty(*x.shape, device=DEV)
    K.bn_eval_fwd(dev(x), dev(gamma), dev(beta), y, dev(rm), dev(rv), swish=True)
    assert_close(y, swish(F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)), 'bn eval')


# ----------------------------------------------------------------------------- small ops
def test_swish_and_embedding():
    x = g(1000, 37, seed=50) * 3
    y = torch.empty(*x.shape, device=DEV)
    K.swish_fwd(dev(x), y)
    assert_close(y, swish(x), 'swish')
    dy = g(*x.shape, seed=51)
    dx = torch.empty(*x.shape, device=DEV)
    K.swish_bwd(dev(dy), dev(x), dx)
    assert_close(dx, dy * swish_grad(x), 'swish bwd')
    for idx in (torch.randint(0, 10, (300,), generator=torch.Generator().manual_seed(52)),
                torch.randint(0, 2, (300,), generator=torch.Generator().manual_seed(53)).float()):
        ncls = 10 if idx.dtype == torch.int64 else 2
        w = g(ncls, 512, seed=54).requires_grad_()
        ref = swish(F.embedding(idx.long(), w))
        da = g(300, 512, seed=55)
        ref.backward(da)
        act = torch.empty(300, 512, device=DEV)
        K.embedding_swish_fwd(dev(idx), dev(w.detach()), act)
        assert_close(act, ref, 'embedding fwd')
        dw = torch.empty(ncls, 512, device=DEV)
        K.embedding_swish_bwd(dev(idx), dev(w.detach()), dev(da), dw)
        assert_close(dw, w.grad, 'embedding bwd')


def test_dropout_fan():
    h = g(40, 512, seed=56)
    masks = (torch.rand(3, 40, 512, generator=torch.Generator().manual_seed(57)) < 0.9).float()
    out = torch.empty(120, 512, device=DEV)
    K.dropout_fanout_fwd(dev(h), dev(masks), out, 1 / 0.9)
    assert_close(out, (h.unsqueeze(0) * (masks / 0.9)).reshape(120, 512), 'fanout')
    dout = g(120, 512, seed=58)
    dh = torch.empty(40, 512, device=DEV)
    K.dropout_fanin_bwd(dev(dout), dev(masks), dh, 1 / 0.9)
    assert_close(dh, (dout.reshape(3, 40, 512) * (masks / 0.9)).sum(0), 'fanin')


# ----------------------------------------------------------------------------- PoE + reparam + KL
@pytest.mark.parametrize('variant,D,B,E,masks', [
    ('A', 64, 37, 2, [0b01, 0b11, 0b10]),
    ('B', 100, 16, 3, [0b101, 0b010, 0b100]),
    ('B', 100, 9, 19, [(1 << 19) - 1, 1, 2, 1 << 18, 0b1010101, 0b110]),
    # celeba19's step: 3 image draws + 18 attributes = 21 experts, 21 terms (complete, image only, 18 single attributes,
    # one sampled subset) -- the launch is cut into blocks of 3 terms / 3 experts (poe.hip, MVAE_POE_CHUNK); expert 2
    # (the sampled term's image draw) is in no term here: its gradient must come out zero
    ('B', 100, 10, 21, [1 | (((1 << 18) - 1) << 3), 2] + [1 << (3 + i) for i in range(18)] + [0b1011 << 5]),
    ('A', 64, 5, 7, [0b1111111, 0b1, 0b10, 0b1000000, 0b0101010]),
])
def test_poe(variant, D, B, E, masks):
    T = len(masks)
    heads = [g(B, 2 * D, seed=60 + e, scale=0.7).requires_grad_() for e in range(E)]
    noise = g(T, B, D, seed=90)
    mus_ref, lvs_ref, zs_ref, kls_ref = [], [], [], []
    for t, m in enumerate(masks):
        sel = [e for e in range(E) if (m >> e) & 1]
        mu, lv = OF.poe_with_prior([heads[e][:, :D] for e in sel], [heads[e][:, D:] for e in sel], variant)
        mus_ref.append(mu); lvs_ref.append(lv)
        zs_ref.append(OF.reparametrize(mu, lv, noise[t])); kls_ref.append(OF.kl_rows(mu, lv))
    mu_r, lv_r, z_r, kl_r = map(torch.stack, (mus_ref, lvs_ref, zs_ref, kls_ref))
    dz, dmu, dlv, dkl = g(T, B, D, seed=91), g(T, B, D, seed=92), g(T, B, D, seed=93), g(T, B, seed=94)
    ((z_r * dz).sum() + (mu_r * dmu).sum() + (lv_r * dlv).sum() + (kl_r * dkl).sum()).backward()

    hd = [dev(h.detach()) for h in heads]
    mus = [h[:, :D] for h in hd]; lvs = [h[:, D:] for h in hd]
    md = torch.tensor(masks, dtype=torch.int32, device=DEV)
    mu = torch.empty(T, B, D, device=DEV); lv = torch.empty_like(mu); z = torch.empty_like(mu)
    kl = torch.empty(T, B, device=DEV)
    K.poe_fwd(mus, lvs, md, dev(noise), mu, lv, z, kl, variant)
    assert_close(mu, mu_r, 'poe mu', tol=1e-5)
    assert_close(lv, lv_r, 'poe logvar', tol=1e-5)
    assert_close(z, z_r, 'poe z', tol=1e-5)
    assert_close(kl, kl_r, 'poe kl', tol=1e-5)
    gh = [torch.empty_like(h) for h in hd]
    K.poe_bwd(mus, lvs, md, dev(noise), mu, lv, dev(dz), dev(dmu), dev(dlv), dev(dkl),
              [x[:, :D] for x in gh], [x[:, D:] for x in gh], variant)
    for e in range(E):      # an expert no term contains gets no gradient from autograd and zeros from the launch
        ref = heads[e].grad if heads[e].grad is not None else torch.zeros_like(heads[e])
        if heads[e].grad is None:
            assert not any((m >> e) & 1 for m in masks) and float(gh[e].abs().max()) == 0.0
        else:
            assert_close(gh[e], ref, 'poe grad expert %d' % e)
    # eval mode: z = mu
    K.poe_fwd(mus, lvs, md, None, mu, lv, z, kl, variant)
    assert_close(z, mu_r, 'poe eval z', tol=1e-5)


def test_poe_per_term_kl_scale():
    B, D, T = 8, 64, 3
    heads = [g(B, 2 * D, seed=70 + e).requires_grad_() for e in range(2)]
    masks = [1, 3, 2]
    coef = torch.tensor([0.1, 0.2, 0.3])
    tot = 0
    for t, m in enumerate(masks):
        sel = [e for e in range(2) if (m >> e) & 1]
        mu, lv = OF.poe_with_prior([heads[e][:, :D] for e in sel], [heads[e][:, D:] for e in sel], 'A')
        tot = tot + coef[t] * OF.kl_rows(mu, lv).sum()
    tot.backward()
    hd = [dev(h.detach()) for h in heads]
    mus = [h[:, :D] for h in hd]; lvs = [h[:, D:] for h in hd]
    md = torch.tensor(masks, dtype=torch.int32, device=DEV)
    mu = torch.empty(T, B, D, device=DEV); lv = torch.empty_like(mu); z = torch.empty_like(mu)
    kl = torch.empty(T, B, device=DEV)
    K.poe_fwd(mus, lvs, md, None, mu, lv, z, kl, 'A')
    gh = [torch.empty_like(h) for h in hd]
    K.poe_bwd(mus, lvs, md, None, mu, lv, None, None, None, dev(coef),
              [x[:, :D] for x in gh], [x[:, D:] for x in gh], 'A', dkl_per_term=True)
    for e in range(2):
        assert_close(gh[e], heads[e].grad, 'kl-only grad expert %d' % e)


def test_poe_bwd_split_equals_accumulated_dz():
    """The latent gradient in two per-decoder buffers (terms {0,1} and {1,2}) gives the bits of one dz they were
    accumulated into, image first."""
    B, D, T, E = 40, 64, 3, 2
    masks = [1, 3, 2]
    hd = [dev(g(B, 2 * D, seed=170 + e, scale=0.7)) for e in range(E)]
    mus = [h[:, :D] for h in hd]; lvs = [h[:, D:] for h in hd]
    md = torch.tensor(masks, dtype=torch.int32, device=DEV)
    noise = dev(g(T, B, D, seed=172))
    mu = torch.empty(T, B, D, device=DEV); lv = torch.empty_like(mu); z = torch.empty_like(mu)
    kl = torch.empty(T, B, device=DEV)
    K.poe_fwd(mus, lvs, md, noise, mu, lv, z, kl, 'A')
    dz_a, dz_b = dev(g(2, B, D, seed=173)), dev(g(2, B, D, seed=174))     # a: terms 0,1; b: terms 1,2
    dz = torch.zeros(T, B, D, device=DEV)
    dz[0:2] += dz_a
    dz[1:3] += dz_b
    coef = dev(torch.tensor([0.1, 0.2, 0.3]))
    ref = [torch.empty_like(h) for h in hd]
    K.poe_bwd(mus, lvs, md, noise, mu, lv, dz, None, None, coef, [x[:, :D] for x in ref], [x[:, D:] for x in ref], 'A',
              dkl_per_term=True)
    out = [torch.empty_like(h) for h in hd]
    K.poe_bwd_split(mus, lvs, md, noise, mu, lv, dz_a, [0, 1, -1], dz_b, [-1, 0, 1], coef,
                    [x[:, :D] for x in out], [x[:, D:] for x in out], 'A', dkl_per_term=True)
    for e in range(E):
        assert torch.equal(out[e], ref[e])
    with pytest.raises(RuntimeError):
        K.poe_bwd_split(mus, lvs, md, noise, mu, lv, dz_a, [0, 1], dz_b, [-1, 0, 1], coef,
                        [x[:, :D] for x in out], [x[:, D:] for x in out], 'A', dkl_per_term=True)


def test_kl_and_reparam_standalone():
    from mvae_amd.functional import ReparamFn, _KlRowsFn
    mu, lv, eps = (g(33, 100, seed=s).requires_grad_(s < 98) for s in (96, 97, 98))
    ref = (OF.reparametrize(mu, lv, eps).pow(2).sum() + (OF.kl_rows(mu, lv) * torch.arange(33.)).sum())
    ref.backward()
    md, ld = dev(mu.detach()).requires_grad_(), dev(lv.detach()).requires_grad_()
    out = ReparamFn.apply(md, ld, dev(eps)).pow(2).sum() + (_KlRowsFn.apply(md, ld) * torch.arange(33., device=DEV)).sum()
    out.backward()
    assert_close(out, ref, 'kl+reparam value', tol=1e-5)
    assert_close(md.grad, mu.grad, 'dmu')
    assert_close(ld.grad, lv.grad, 'dlogvar')


# ----------------------------------------------------------------------------- losses
@pytest.mark.parametrize('R,P,groups,colw', [(12, 784, 2, False), (9, 12288, 3, False), (30, 18, 3, True),
                                             (8, 1, 1, False), (6, 600, 1, True)])
def test_bce_rowsum(R, P, groups, colw):
    rpg = R // groups
    x = (g(R, P, seed=100) * 3).requires_grad_()
    t = torch.rand(rpg, P, generator=torch.Generator().manual_seed(101))
    w = torch.rand(groups, P, generator=torch.Generator().manual_seed(102)) if colw else None
    drow = torch.tensor([0.5, 0.0, 2.0][:groups])
    el = OF.binary_cross_entropy_with_logits(x, t.repeat(groups, 1))
    if colw:
        el = el * w.repeat_interleave(rpg, 0)
    rows_ref = el.sum(1)
    (rows_ref * drow.repeat_interleave(rpg)).sum().backward()
    rows = torch.empty(R, device=DEV); dl = torch.empty(R, P, device=DEV)
    K.bce_rowsum_fwd(dev(x.detach()), dev(t), rows, colw=dev(w), drow=dev(drow), dlogits=dl,
                     rows_per_group=rpg, target_rows=rpg)
    assert_close(rows, rows_ref, 'bce rows', tol=1e-5)
    assert_close(dl, x.grad, 'bce fused grad')
    dl2 = torch.empty(R, P, device=DEV)
    K.bce_rowsum_bwd(dev(x.detach()), dev(t), dev(drow), dl2, colw=dev(w), rows_per_group=rpg, target_rows=rpg)
    assert_close(dl2, x.grad, 'bce bwd')


def test_bce_matches_reference_subgradient_at_zero():
    x = torch.tensor([[0.0, 0.0, -0.0, 1.5]], requires_grad=True)
    t = torch.tensor([[0.25, 1.0, 0.0, 0.5]])
    OF.binary_cross_entropy_with_logits(x, t).sum().backward()
    dl = torch.empty(1, 4, device=DEV); rows = torch.empty(1, device=DEV)
    K.bce_rowsum_fwd(dev(x.detach()), dev(t), rows, drow=torch.ones(1, device=DEV), dlogits=dl, rows_per_group=1)
    assert torch.allclose(dl.cpu(), x.grad, atol=1e-7)


def test_cross_entropy_rows():
    R, Kc, groups = 24, 10, 2
    x = (g(R, Kc, seed=103) * 2).requires_grad_()
    y = torch.randint(0, Kc, (R // groups,), generator=torch.Generator().manual_seed(104))
    drow = torch.tensor([0.7, 1.3])
    rows_ref = OF.cross_entropy(x, y.repeat(groups)).sum(1)
    (rows_ref * drow.repeat_interleave(R // groups)).sum().backward()
    rows = torch.empty(R, device=DEV); dl = torch.empty(R, Kc, device=DEV)
    K.ce_fwd(dev(x.detach()), dev(y), rows, drow=dev(drow), dlogits=dl, rows_per_group=R // groups,
             label_rows=R // groups)
    assert_close(rows, rows_ref, 'ce rows', tol=1e-5)
    assert_close(dl, x.grad, 'ce grad')


# the last Linear of a decoder with its reconstruction term in the same launch (mnist/model.py:104,146 ->
# mnist/train.py:47-52): against Linear -> term on the CPU, and bit for bit against the two-launch HIP route on the
# logits the fused launch itself reports
FOLD_SHAPES = [(1536, 784, 512, 3), (24, 784, 512, 3), (96, 18, 512, 3), (1024, 784, 512, 2), (74, 50, 20, 2),
               (3, 5, 7, 1), (512, 33, 64, 1)]


@pytest.mark.parametrize('M,N,Kd,groups', FOLD_SHAPES)
@pytest.mark.parametrize('bias', [True, False])
def test_linear_with_bernoulli_term(M, N, Kd, groups, bias):
    rpg = M // groups
    x = g(M, Kd, seed=140)
    w = (g(N, Kd, seed=141) / Kd ** 0.5).requires_grad_()
    b = (g(N, seed=142) * 0.1).requires_grad_() if bias else None
    t = torch.rand(rpg, N, generator=torch.Generator().manual_seed(143))
    drow = torch.tensor([0.5, 0.0, 2.0][:groups])
    xr = x.clone().requires_grad_()
    logits_ref = F.linear(xr, w, b)
    logits_ref.retain_grad()
    rows_ref = OF.binary_cross_entropy_with_logits(logits_ref, t.repeat(groups, 1)).sum(1)
    (rows_ref * drow.repeat_interleave(rpg)).sum().backward()
    nparts = K.bce_partials(N)
    dl = torch.empty(M, N, device=DEV); lg = torch.empty(M, N, device=DEV)
    part = torch.full((M * nparts,), float('nan'), device=DEV)
    xd, wd, bd, td, dd = dev(x), dev(w.detach()), dev(None if b is None else b.detach()), dev(t), dev(drow)
    K.linear_bce_fwd(xd, wd, bd, td, dd, dl, part, rpg, rpg, logits=lg)
    assert_close(lg, logits_ref.detach(), 'folded logits', tol=1e-5)
    assert_close(part.reshape(M, nparts).sum(1), rows_ref.detach(), 'folded bce rows', tol=1e-5)
    assert_close(dl, logits_ref.grad, 'folded d loss / d logits')
    # the two-launch route on the same logits: the gradient bit for bit, the row sums to summation order
    rows2 = torch.empty(M, device=DEV); dl2 = torch.empty(M, N, device=DEV)
    K.bce_rowsum_fwd(lg, td, rows2, drow=dd, dlogits=dl2, rows_per_group=rpg, target_rows=rpg)
    assert torch.equal(dl, dl2)
    assert_close(part.reshape(M, nparts).sum(1), rows2, 'folded vs kernel rows', tol=1e-6)
    # without the logits output (the step's form): the same bits
    dl3 = torch.empty(M, N, device=DEV); part3 = torch.empty(M * nparts, device=DEV)
    K.linear_bce_fwd(xd, wd, bd, td, dd, dl3, part3, rpg, rpg)
    assert torch.equal(dl, dl3) and torch.equal(part, part3)


@pytest.mark.parametrize('M,N,Kd,groups', [(1536, 10, 512, 3), (24, 10, 512, 2), (70, 32, 20, 2), (5, 1, 7, 1)])
def test_linear_with_categorical_term(M, N, Kd, groups):
    rpg = M // groups
    x = g(M, Kd, seed=150)
    w = (g(N, Kd, seed=151) / Kd ** 0.5 * 3).requires_grad_()
    b = (g(N, seed=152) * 0.1).requires_grad_()
    y = torch.randint(0, N, (rpg,), generator=torch.Generator().manual_seed(153))
    drow = torch.tensor([0.7, 1.3, 0.0][:groups])
    logits_ref = F.linear(x, w, b)
    logits_ref.retain_grad()
    rows_ref = OF.cross_entropy(logits_ref, y.repeat(groups)).sum(1)
    (rows_ref * drow.repeat_interleave(rpg)).sum().backward()
    dl = torch.empty(M, N, device=DEV); lg = torch.empty(M, N, device=DEV)
    rows = torch.full((M,), float('nan'), device=DEV)
    K.linear_ce_fwd(dev(x), dev(w.detach()), dev(b.detach()), dev(y), dev(drow), dl, rows, rpg, rpg, logits=lg)
    assert_close(lg, logits_ref.detach(), 'folded logits', tol=1e-5)
    assert_close(rows, rows_ref.detach(), 'folded ce rows', tol=1e-5)
    assert_close(dl, logits_ref.grad, 'folded ce grad')
    rows2 = torch.empty(M, device=DEV); dl2 = torch.empty(M, N, device=DEV)
    K.ce_fwd(lg, dev(y), rows2, drow=dev(drow), dlogits=dl2, rows_per_group=rpg, label_rows=rpg)
    assert_close(rows, rows2, 'folded vs kernel ce rows', tol=1e-6)
    assert_close(dl, dl2, 'folded vs kernel ce grad', tol=1e-6)


def test_linear_with_categorical_term_flags_a_bad_label():
    M, N, Kd = 8, 10, 16
    y = torch.tensor([0, 3, 10, 9, -1, 2, 5, 7])
    dl = torch.empty(M, N, device=DEV); rows = torch.empty(M, device=DEV)
    K.linear_ce_fwd(dev(g(M, Kd, seed=154)), dev(g(N, Kd, seed=155)), None, dev(y), torch.ones(1, device=DEV), dl, rows,
                    M, M)
    bad = torch.tensor([False, False, True, False, True, False, False, False])
    assert torch.equal(torch.isnan(rows).cpu(), bad)
    assert torch.equal(torch.isnan(dl).all(1).cpu(), bad) and not torch.isnan(dl[~bad.to(DEV)]).any()


def test_group_sums():
    rows = g(3 * 50, seed=105)
    coef = torch.tensor([0.5, 0.0, 2.0])
    out = torch.zeros(3, device=DEV); tot = torch.zeros(1, device=DEV)
    K.group_sums(dev(rows), dev(coef), out, tot, 3, 50)
    ref = rows.reshape(3, 50).sum(1) * coef
    assert_close(out, ref, 'group sums', tol=1e-5)
    assert_close(tot, ref.sum().reshape(1), 'group total', tol=1e-5)
    K.group_sums(dev(rows), dev(coef), out, tot, 3, 50, accumulate=True)
    assert_close(tot, 2 * ref.sum().reshape(1), 'group total accumulate', tol=1e-5)


# ----------------------------------------------------------------------------- noise / Adam
def test_philox_noise_statistics_and_counter():
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    a = torch.empty(1 << 20, device=DEV); b = torch.empty(1 << 20, device=DEV)
    K.randn_(a, 1234, ctr); K.randn_(b, 1234, ctr)
    assert ctr.item() == 2
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs((a * a * a * a).mean().item() - 3) < 0.05          # kurtosis of N(0,1)
    assert (a != b).float().mean().item() > 0.999                  # a new launch is a new stream
    ctr.zero_()
    c = torch.empty(1 << 20, device=DEV)
    K.randn_(c, 1234, ctr)
    assert torch.equal(a, c)                                        # counter-based: reproducible
    m = torch.empty(1 << 20, device=DEV)
    K.bernoulli_(m, 0.9, 99, ctr)
    assert abs(m.mean().item() - 0.9) < 2e-3 and set(m.unique().tolist()) == {0.0, 1.0}


def test_fused_adam_matches_torch_adam():
    n = 10007
    p0, grads = g(n, seed=110), [g(n, seed=111 + i) for i in range(5)]
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-3)
    p = dev(p0).clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    for gi in grads:
        pr.grad = gi.clone(); opt.step()
        K.adam_step(p, dev(gi * 4), m, v, step, 1e-3, grad_scale=0.25)
    assert step.item() == 5
    assert_close(p, pr.detach(), 'adam params', tol=1e-6)
    assert_close(m, opt.state[pr]['exp_avg'], 'adam m', tol=1e-6)
    assert_close(v, opt.state[pr]['exp_avg_sq'], 'adam v', tol=1e-6)


def test_adam_apply_over_ranges_equals_one_step():
    """Data-parallel replicas run Adam per gradient bucket (mvae_adam_apply: no counter advance) and advance
    the step once: identical to one mvae_adam_step over the whole arena."""
    n = 10008
    p0, m0, v0, gr = g(n, seed=120), g(n, seed=121).abs() * 0.1, g(n, seed=122).abs() * 0.1, g(n, seed=123)
    step_a = torch.full((1,), 3, dtype=torch.int64, device=DEV); step_b = step_a.clone()
    pa, ma, va = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
    pb, mb, vb = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
    K.adam_step(pa, dev(gr), ma, va, step_a, 1e-3, grad_scale=0.5)
    gd = dev(gr)
    for lo, hi in ((0, 4000), (4000, 9000), (9000, n)):
        K.adam_apply(pb[lo:hi], gd[lo:hi], mb[lo:hi], vb[lo:hi], step_b, 1e-3, grad_scale=0.5)
    assert step_b.item() == 3
    K.counter_add(step_b, 1)
    assert step_b.item() == step_a.item() == 4
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


def test_adam_apply_at_with_the_counter_advanced_first_equals_one_step():
    """The captured single-GPU step advances the counter early (off the critical chain) and updates with
    mvae_adam_apply_at(step_add=0): identical to mvae_adam_step, counter included."""
    n = 10009
    p0, m0, v0, gr = g(n, seed=124), g(n, seed=125).abs() * 0.1, g(n, seed=126).abs() * 0.1, g(n, seed=127)
    step_a = torch.full((1,), 6, dtype=torch.int64, device=DEV); step_b = step_a.clone()
    pa, ma, va = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
    pb, mb, vb = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
    K.adam_step(pa, dev(gr), ma, va, step_a, 1e-3, grad_scale=0.5)
    K.counter_add(step_b, 1)
    K.adam_apply_at(pb, dev(gr), mb, vb, step_b, 0, 1e-3, grad_scale=0.5)
    assert step_b.item() == step_a.item() == 7
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    pc, mc, vc = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
    step_c = torch.full((1,), 6, dtype=torch.int64, device=DEV)
    K.adam_apply_at(pc, dev(gr), mc, vc, step_c, 1, 1e-3, grad_scale=0.5)        # step_add = 1: mvae_adam_apply
    assert step_c.item() == 6 and torch.equal(pa, pc)


def test_philox_fill_matches_the_bumping_entry_points():
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    a = torch.empty(4099, device=DEV); m = torch.empty(4099, device=DEV)
    K.randn_(a, 77, ctr); K.bernoulli_(m, 0.9, 78, ctr)          # launch indices 0 and 1, counter -> 2
    ctr2 = torch.zeros(1, dtype=torch.int64, device=DEV)
    a2 = torch.empty_like(a); m2 = torch.empty_like(m)
    K.philox_fill(a2, 77, ctr2, 0)
    K.philox_fill(m2, 78, ctr2, 1, keep_prob=0.9)
    assert ctr2.item() == 0 and torch.equal(a, a2) and torch.equal(m, m2)


def test_elbo_reduce_equals_the_separate_launches():
    T, B = 5, 37
    kl, ri, rl = g(T * B, seed=130), g(2 * B, seed=131), g(3 * B, seed=132)
    ck, ci, cl = torch.rand(T), torch.rand(2), torch.rand(3)
    vals, cv = g(9, seed=133), torch.rand(9)
    term_of = torch.tensor([0, 4, 4, 1, 2, 0, 3, 3, 1], dtype=torch.int32)
    # the separate launches the fused step used to issue
    out = torch.zeros(T + 1, device=DEV)
    K.group_sums(dev(kl), dev(ck), out[:T], out[T:], T, B, accumulate=False)
    K.group_sums(dev(ri), dev(ci), out[1:3], out[T:], 2, B, accumulate=True)
    K.group_sums(dev(rl), dev(cl), out[2:5], out[T:], 3, B, accumulate=True)
    K.scatter_sums(dev(vals), dev(cv), dev(term_of), out[:T], out[T:], accumulate_total=True)
    fused = torch.full((T + 1,), 7.0, device=DEV)
    zero = torch.full((4099,), 3.0, device=DEV)
    ctr = torch.full((1,), 5, dtype=torch.int64, device=DEV)
    K.elbo_reduce([(dev(kl), dev(ck), None, 0, T, B), (dev(ri), dev(ci), None, 1, 2, B), (dev(rl), dev(cl), None, 2, 3, B),
                   (dev(vals), dev(cv), dev(term_of), 0, 9, 1)], fused, T, zero=zero, counter_dev=ctr, counter_inc=2)
    assert_close(fused, out, 'fused ELBO vs separate launches', tol=1e-6)
    ref = torch.zeros(T + 1, dtype=torch.float64)
    ref[:T] += (kl.double().reshape(T, B).sum(1) * ck.double())
    ref[1:3] += ri.double().reshape(2, B).sum(1) * ci.double()
    ref[2:5] += rl.double().reshape(3, B).sum(1) * cl.double()
    for j in range(9):
        ref[term_of[j]] += cv[j].double() * vals[j].double()
    ref[T] = ref[:T].sum()
    assert_close(fused, ref, 'fused ELBO vs fp64', tol=1e-5)
    assert zero.abs().max().item() == 0 and ctr.item() == 7
    with pytest.raises(RuntimeError):
        K.elbo_reduce([(dev(kl), dev(ck), None, 3, T, B)], fused, T)          # terms 3..7 do not fit T = 5


def test_sigmoid_and_affine():
    x = g(3, 17, seed=140)
    y = torch.empty(3, 17, device=DEV)
    K.sigmoid_fwd(dev(x), y)
    assert_close(y, torch.sigmoid(x), 'sigmoid', tol=1e-6)
    sc, sh = g(17, seed=141), g(17, seed=142)
    K.affine_fwd(dev(x), dev(sc), dev(sh), y)
    assert_close(y, x * sc + sh, 'affine (row broadcast)', tol=1e-6)
    K.affine_fwd(dev(x), torch.ones(1, device=DEV), torch.zeros(1, device=DEV), y)
    assert torch.equal(y.cpu(), x)


def test_ingest_one_launch_for_batch_and_tables():
    """mvae_ingest: image + label batch to their static buffers and the PINNED host table block to its device block
    (read by the kernel itself), bit-exact, for int64 labels and float attribute rows."""
    g = torch.Generator().manual_seed(3)
    for label in (torch.randint(0, 10, (33,), generator=g), torch.randint(0, 2, (33, 18), generator=g).float()):
        image = torch.rand(33, 1, 28, 28, generator=g).to(DEV)
        label = label.to(DEV)
        table = torch.arange(57, dtype=torch.int32).pin_memory()
        s_img, s_lbl = torch.zeros_like(image), torch.zeros_like(label)
        s_tbl = torch.zeros(57, dtype=torch.int32, device=DEV)
        assert K.ingest_ok(image, s_img, label, s_lbl)
        K.ingest(image, s_img, label, s_lbl, table, s_tbl)
        torch.cuda.synchronize()
        assert torch.equal(s_img, image) and torch.equal(s_lbl, label) and torch.equal(s_tbl.cpu(), table)
    assert not K.ingest_ok(image.cpu(), s_img, label, s_lbl)
    assert not K.ingest_ok(image[:, :, :, :27], s_img[:, :, :, :27], label, s_lbl)       # not contiguous
    with pytest.raises(RuntimeError, match='pinned'):
        K.ingest(image, s_img, label, s_lbl, torch.arange(57, dtype=torch.int32), s_tbl)


# ----------------------------------------------------------------------------- statistics-only transposed conv
@pytest.mark.parametrize('G,B,Cin,H,Cout', [(3, 8, 64, 16, 32), (1, 2, 64, 16, 32), (18, 16, 64, 16, 32), (2, 4, 16, 8, 8)])
def test_convT_stats_only_matches_conv_then_batchnorm(G, B, Cin, H, Cout):
    """mvae_convT2d_k4_fwd_stats + mvae_bn_stats_merge (the last layer of a decoder pass that exists only for its
    BatchNorm running statistics: nothing stored) == ConvTranspose2d then training-mode BatchNorm statistics, per group,
    n_updates times each: 1e-5 on the running statistics, like the storing launch + statistics sweep it replaces."""
    x = g(G * B, Cin, H, H, seed=60) * 0.8 + 0.3
    w = g(Cin, Cout, 4, 4, seed=61, scale=(Cin * 4) ** -0.5)
    y = F.conv_transpose2d(x, w, None, 2, 1)
    rm, rv = 0.05 * g(Cout, seed=62), 1 + 0.1 * torch.rand(Cout, generator=torch.Generator().manual_seed(63))
    rm0, rv0 = rm.clone(), rv.clone()
    means, invstds = [], []
    for gi in range(G):
        yg = y[gi * B:(gi + 1) * B]
        for _ in range(2):
            F.batch_norm(yg, rm, rv, None, None, True, 0.1, 1e-5)
        means.append(yg.mean(dim=(0, 2, 3)))
        invstds.append((yg.var(dim=(0, 2, 3), unbiased=False) + 1e-5).rsqrt())
    xd, wd = dev(x), dev(w)
    tiles = K.convT2d_stats_tiles(xd, wd, 2, 1)
    assert tiles == G * B * H * H // 128 and tiles % G == 0
    part = K.convT2d_fwd_stats(xd, wd, 2, 1)
    assert part.shape == (tiles, Cout, 2)
    sm = torch.empty(G, Cout, device=DEV); si = torch.empty(G, Cout, device=DEV)
    rmd, rvd = dev(rm0), dev(rv0)
    K.bn_stats_merge(part, G, sm, si, rmd, rvd, n_updates=2)
    assert_close(sm, torch.stack(means), 'group means', tol=1e-5)
    assert_close(si, torch.stack(invstds), 'group invstd', tol=1e-5)
    assert_close(rmd, rm, 'running_mean', tol=1e-5)
    assert_close(rvd, rv, 'running_var', tol=1e-5)
    # ... and the route it replaces: the storing launch, then the statistics-only BatchNorm sweep
    pre = torch.empty(*y.shape, device=DEV)
    K.convT2d_fwd(xd, wd, pre, None, 2, 1)
    rm2, rv2 = dev(rm0), dev(rv0)
    sm2 = torch.empty(G, Cout, device=DEV); si2 = torch.empty(G, Cout, device=DEV)
    K.bn_train_fwd(pre, torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV), None, sm2, si2, rm2, rv2, G,
                   n_updates=2, swish=False)
    assert_close(rmd, rm2, 'running_mean vs sweep', tol=1e-5)
    assert_close(rvd, rv2, 'running_var vs sweep', tol=1e-5)


def test_convT_stats_only_with_a_channel_mean_far_from_zero():
    """the statistics-only epilogue took M2 as sum(v^2) - sum(v) * mean in fp32 -- at |mean| / std ~ 1e3 every
    digit of the variance cancels (and a clamp at 0 hid it), while the two-pass BatchNorm kernels it replaces have no such
    limit.  Sums are now taken around a sample of the row.  Here every output sits near 160 with a spread of ~0.1 (only
    the four centre taps carry weight: with stride 2 / pad 1 each output then has exactly one tap per dimension, borders too)."""
    G, B, Cin, H, Cout = 1, 8, 64, 16, 32
    x = g(G * B, Cin, H, H, seed=64).abs() * 0.01 + 5.0
    w = torch.zeros(Cin, Cout, 4, 4)
    w[:, :, 1:3, 1:3] = 0.5 + 0.002 * g(Cin, Cout, 2, 2, seed=65)
    y = F.conv_transpose2d(x.double(), w.double(), None, 2, 1)
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    assert (mean.abs() / var.sqrt()).min().item() > 1e3           # the hard regime
    xd, wd = dev(x), dev(w)
    part = K.convT2d_fwd_stats(xd, wd, 2, 1)
    sm = torch.empty(G, Cout, device=DEV); si = torch.empty(G, Cout, device=DEV)
    rmd, rvd = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    K.bn_stats_merge(part, G, sm, si, rmd, rvd, n_updates=1)
    n = y.numel() // Cout
    rv_ref = 0.9 + 0.1 * var * n / (n - 1)
    assert_close(sm[0], mean.float(), 'mean', tol=1e-6)
    # the conv's own fp32 round-off (~1e-5 absolute on values of 160) bounds what any variance estimate can reach here
    assert_close(si[0], (var + 1e-5).rsqrt().float(), 'invstd at mean/std > 1e3', tol=2e-2)
    assert_close(rvd, rv_ref.float(), 'running_var at mean/std > 1e3', tol=2e-2)
    # ... and the storing launch + two-pass sweep agree with it as closely
    pre = torch.empty(G * B, Cout, 2 * H, 2 * H, device=DEV)
    K.convT2d_fwd(xd, wd, pre, None, 2, 1)
    rm2, rv2 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    sm2 = torch.empty(G, Cout, device=DEV); si2 = torch.empty(G, Cout, device=DEV)
    K.bn_train_fwd(pre, torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV), None, sm2, si2, rm2, rv2, G,
                   n_updates=1, swish=False)
    assert_close(rvd, rv2, 'running_var vs the two-pass sweep', tol=2e-2)


def test_convT_stats_only_refuses_what_it_does_not_cover():
    xd = torch.zeros(4, 64, 16, 16, device=DEV)
    assert K.convT2d_stats_tiles(xd, torch.zeros(64, 64, 4, 4, device=DEV), 2, 1) == 0       # 64 output channels
    assert K.convT2d_stats_tiles(torch.zeros(3, 64, 5, 5, device=DEV), torch.zeros(64, 32, 4, 4, device=DEV), 2, 1) == 0
    assert K.convT2d_stats_tiles(torch.zeros(4, 256, 5, 5, device=DEV), torch.zeros(256, 32, 4, 4, device=DEV), 1, 0) == 0
    with pytest.raises(RuntimeError):
        K.convT2d_fwd_stats(xd, torch.zeros(64, 64, 4, 4, device=DEV), 2, 1)

"""CPU: the control flow of layers.forward_tape with statistics launches (MVAE_FUSED_BN_STATS=1, experimental) and
the record layout as the header specifies it.  The HIP entry points are replaced by torch stand-ins that follow
include/mvae_hip.h to the letter -- producer: one (mean, M2) record per channel and `cols` consecutive output
positions at stats[(part * 2 + {0,1}) * C + c], part = (class * tiles_j + tile) * ppt + wave, positions numbered
(image, row', col') over the class's sub-lattice; consumer: a group's records are tiles [g * tiles_j / G, ...) of
every class -- while the layout itself comes from the real library (mvae_conv_k4_stats_layout is host logic).
What this pins: which launches forward_tape issues, what it stores, that a stats_only pass stores nothing, and
that the layout arithmetic of producer and consumer agree for one group and several.  The device code is covered
by tests/test_fused_bn_stats_gpu.py."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import mvae_amd
from mvae_amd import kernels as K, layers as L


def swish(x):
    return x * torch.sigmoid(x)


class Calls(list):
    pass


def install_standins(monkeypatch, calls):
    def conv2d_fwd(x, w, pre, act, s, p):
        calls.append('conv2d_fwd')
        y = F.conv2d(x, w, None, s, p)
        if pre is not None: pre.copy_(y)
        if act is not None: act.copy_(swish(y))

    def convT2d_fwd(x, w, pre, act, s, p, wr=None):
        calls.append('convT2d_fwd')
        y = F.conv_transpose2d(x, w, None, s, p)
        if pre is not None: pre.copy_(y)
        if act is not None: act.copy_(swish(y))

    def records(y, lay, s_cls, stats):
        N, C = y.shape[0], y.shape[1]
        part = 0
        for cls in range(lay.ncls):
            py, px = (cls // s_cls, cls % s_cls) if lay.ncls > 1 else (0, 0)
            sub = y[:, :, py::s_cls, px::s_cls] if lay.ncls > 1 else y
            cols = sub.permute(1, 0, 2, 3).reshape(C, -1)                  # [C, J], j = (image, row', col')
            assert cols.shape[1] == lay.tiles_j * lay.ppt * lay.cols
            chunks = cols.reshape(C, lay.tiles_j * lay.ppt, lay.cols).double()
            mean = chunks.mean(2); m2 = ((chunks - mean[..., None]) ** 2).sum(2)
            n = lay.tiles_j * lay.ppt
            view = stats[:lay.parts() * 2 * C].reshape(lay.parts(), 2, C)
            view[part:part + n, 0] = mean.t().float()
            view[part:part + n, 1] = m2.t().float()
            part += n

    def conv2d_fwd_stats(x, w, pre, s, p, stats):
        calls.append('conv2d_fwd_stats' + ('' if pre is not None else ':nostore'))
        y = F.conv2d(x, w, None, s, p)
        lay = K.conv_stats_layout(0, x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0], s, p)
        records(y, lay, 1, stats)
        if pre is not None: pre.copy_(y)
        return lay

    def convT2d_fwd_stats(x, w, pre, s, p, stats, wr=None):
        calls.append('convT2d_fwd_stats' + ('' if pre is not None else ':nostore'))
        y = F.conv_transpose2d(x, w, None, s, p)
        lay = K.conv_stats_layout(1, x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[1], s, p)
        records(y, lay, s, stats)
        if pre is not None: pre.copy_(y)
        return lay

    def finish_bn(x, mean, var, n, gamma, beta, y, sm, si, rm, rv, G, eps, momentum, n_updates, sw):
        sm.copy_(mean.float()); si.copy_((var + eps).rsqrt().float())
        for g in range(G):
            for _ in range(n_updates):
                rm.mul_(1 - momentum).add_(momentum * mean[g].float())
                rv.mul_(1 - momentum).add_(momentum * (var[g] * n / (n - 1)).float())
        if y is not None:
            B = x.shape[0] // G
            xs = x.reshape(G, B, x.shape[1], -1).double()
            h = (xs - mean[:, None, :, None]) * (var + eps).rsqrt()[:, None, :, None] * gamma.double()[None, None, :, None] \
                + beta.double()[None, None, :, None]
            y.copy_((swish(h) if sw else h).float().reshape(y.shape))

    def bn_train_fwd(x, gamma, beta, y, sm, si, rm, rv, G, eps=1e-5, momentum=0.1, n_updates=1, swish=True,
                     n_updates_dev=None):
        calls.append('bn_train_fwd' + ('' if y is not None else ':stats_only'))
        B = x.shape[0] // G
        xs = x.reshape(G, B, x.shape[1], -1).double()
        finish_bn(x, xs.mean(dim=(1, 3)), xs.var(dim=(1, 3), unbiased=False), B * xs.shape[3], gamma, beta, y, sm, si,
                  rm, rv, G, eps, momentum, n_updates, swish)

    def bn_train_fwd_parts(x, gamma, beta, y, sm, si, rm, rv, G, shape, stats, lay, eps=1e-5, momentum=0.1,
                           n_updates=1, swish=True, n_updates_dev=None):
        calls.append('bn_train_fwd_parts' + ('' if y is not None else ':stats_only'))
        assert (x is None) == (y is None) or x is not None
        C = shape[1]
        assert lay.tiles_j % G == 0 and lay.parts() * lay.cols == shape[0] * shape[2] * shape[3]
        tpg = lay.tiles_j // G
        view = stats[:lay.parts() * 2 * C].reshape(lay.ncls, lay.tiles_j, lay.ppt, 2, C).double()
        mean = torch.empty(G, C, dtype=torch.float64); var = torch.empty(G, C, dtype=torch.float64)
        for g in range(G):
            r = view[:, g * tpg:(g + 1) * tpg].reshape(-1, 2, C)
            m = r[:, 0].mean(0)                                       # equal counts
            m2 = r[:, 1].sum(0) + lay.cols * ((r[:, 0] - m) ** 2).sum(0)
            mean[g] = m; var[g] = m2 / (r.shape[0] * lay.cols)
        n = shape[0] // G * shape[2] * shape[3]
        finish_bn(x, mean, var, n, gamma, beta, y, sm, si, rm, rv, G, eps, momentum, n_updates, swish)

    for name, fn in (('conv2d_fwd', conv2d_fwd), ('convT2d_fwd', convT2d_fwd), ('conv2d_fwd_stats', conv2d_fwd_stats),
                     ('convT2d_fwd_stats', convT2d_fwd_stats), ('bn_train_fwd', bn_train_fwd),
                     ('bn_train_fwd_parts', bn_train_fwd_parts)):
        monkeypatch.setattr(K, name, fn)
    monkeypatch.setattr(L, '_probe_repack', lambda *a, **k: None)
    monkeypatch.setattr(L, '_fresh_repack', lambda m: None)


def encoder():      # celeba/model.py:77-86 without the first (3-channel) conv
    return [L.Conv2d(32, 64, 4, 2, 1, bias=False), L.BatchNorm2d(64), L.Swish(),
            L.Conv2d(64, 128, 4, 2, 1, bias=False), L.BatchNorm2d(128), L.Swish()]


def decoder():      # celeba/model.py:120-126
    return [L.ConvTranspose2d(128, 64, 4, 2, 1, bias=False), L.BatchNorm2d(64), L.Swish(),
            L.ConvTranspose2d(64, 32, 4, 2, 1, bias=False), L.BatchNorm2d(32), L.Swish(),
            L.ConvTranspose2d(32, 3, 4, 2, 1, bias=False)]


def run(mods, x, groups, flag, monkeypatch, stats_only=False):
    torch.manual_seed(3)
    mods = mods()
    for m in mods:
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.2, 0.2)
    plan = L.compile_plan(mods)
    calls = Calls()
    install_standins(monkeypatch, calls)
    monkeypatch.setenv('MVAE_FUSED_BN_STATS', flag)
    out, tape = L.forward_tape(plan, x, groups=groups, bn_updates=2, stats_only=stats_only)
    bns = [m for m in mods if isinstance(m, nn.BatchNorm2d)]
    return out, tape, calls, [(m.running_mean.clone(), m.running_var.clone(), m._nbt_pending) for m in bns]


@pytest.mark.parametrize('mods,shape,groups', [(encoder, (8, 32, 32, 32), 1), (decoder, (3 * 8, 128, 4, 4), 3),
                                               (decoder, (2 * 16, 128, 4, 4), 2)])
def test_forward_tape_with_statistics_launches_equals_the_sweep(mods, shape, groups, monkeypatch):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1)) + 0.2
    ya, ta, ca, ra = run(mods, x, groups, '0', monkeypatch)
    yb, tb, cb, rb = run(mods, x, groups, '1', monkeypatch)
    assert not any('stats' in c for c in ca if c.startswith('conv'))
    n_bn = sum(1 for c in ca if c.startswith('bn_train_fwd'))
    assert sum(1 for c in cb if c.startswith('bn_train_fwd_parts')) == n_bn, cb       # every BatchNorm fed by records
    assert not any(c == 'bn_train_fwd' for c in cb)
    assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-6)
    for (ma, va, na), (mb, vb, nb) in zip(ra, rb):
        assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-7) and torch.allclose(va, vb, rtol=1e-5, atol=1e-7) and na == nb
    assert len(ta) == len(tb)
    for sa, sb in zip(ta, tb):                                  # same tape: what the backward reads
        assert len(sa) == len(sb)
        for a, b in zip(sa, sb):
            assert (a is None) == (b is None)
            if a is not None:
                assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_stats_only_pass_stores_nothing_with_statistics_launches(monkeypatch):
    x = torch.randn(3 * 8, 128, 4, 4, generator=torch.Generator().manual_seed(2))
    ya, ta, ca, ra = run(decoder, x, 3, '0', monkeypatch, stats_only=True)
    yb, tb, cb, rb = run(decoder, x, 3, '1', monkeypatch, stats_only=True)
    assert ya is None and yb is None and ta is None and tb is None
    assert ca == ['convT2d_fwd', 'bn_train_fwd', 'convT2d_fwd', 'bn_train_fwd:stats_only']
    assert cb == ['convT2d_fwd_stats', 'bn_train_fwd_parts', 'convT2d_fwd_stats:nostore', 'bn_train_fwd_parts:stats_only']
    for (ma, va, na), (mb, vb, nb) in zip(ra, rb):
        assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-7) and torch.allclose(va, vb, rtol=1e-5, atol=1e-7) and na == nb


def test_shapes_without_a_statistics_launch_fall_back(monkeypatch):
    x = torch.randn(7, 32, 32, 32, generator=torch.Generator().manual_seed(4))
    mods = lambda: [L.Conv2d(32, 64, 4, 2, 1, bias=False), L.BatchNorm2d(64), L.Swish(),
                    L.Conv2d(64, 128, 4, 2, 1, bias=False), L.BatchNorm2d(128), L.Swish(),
                    L.Conv2d(128, 256, 4, 1, 0, bias=False), L.BatchNorm2d(256), L.Swish()]
    y, t, calls, _ = run(mods, x, 1, '1', monkeypatch)
    # 7 images: 7*256 and 7*64 positions are whole 64-wide tiles, 7*25 are not
    assert calls == ['conv2d_fwd_stats', 'bn_train_fwd_parts', 'conv2d_fwd_stats', 'bn_train_fwd_parts',
                     'conv2d_fwd', 'bn_train_fwd']

"""GPU: the data-parallel launch path (three captured graphs with the RCCL bucket all-reduces issued
between them) on ONE GPU, world_size 1 over the nccl (= RCCL) backend -- the same code the N > 1
bench runs, minus peers.  Checks that it trains identically to the single-graph path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import mvae_amd
from mvae_amd.engine import BimodalStep
from mvae_amd.optim import FusedAdam
from mvae_amd.parallel import DataParallel
from oracle import steps as OS
from test_engine_gpu import build_pair
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture
def nccl_world1():
    from util import init_world1
    init_world1('nccl', torch.device('cuda', 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize('kind,batch', [('mnist', 32), ('celeba', 8)])
def test_dp_three_graph_path_matches_single_graph(nccl_world1, kind, batch, monkeypatch):
    monkeypatch.setenv('MVAE_COMM', 'torch')      # the torch.distributed transport: one graph per bucket
    lam = 50.0 if kind == 'mnist' else 10.0
    runs = []
    for use_dp in (False, True):
        _, model, d = build_pair(kind, weight_seed=41)
        eng = BimodalStep(model, batch, 1.0, lam, seed=7)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        dp = DataParallel(model, eng) if use_dp else None
        image, label = OS.synthetic_batch(kind, batch, seed=90)
        eng.capture(opt, image.shape[1:], label, comm=dp)
        losses = []
        for step in range(3):
            image, label = OS.synthetic_batch(kind, batch, seed=91 + step)
            losses.append(eng.replay(image.to(DEV), label.to(DEV), 0.25 * (step + 1))[-1].item())
        torch.cuda.synchronize()
        runs.append((losses, model.arena.flat.clone()))
        if use_dp:
            # one graph per gradient bucket: MNIST keeps its 4 MB of encoders in one, CelebA splits off
            # the conv trunk (arena tail) as a small last bucket
            assert len(eng._graphs) == dp.n_buckets == (2 if kind == 'mnist' else 3)
    assert_close(torch.tensor(runs[1][0]), torch.tensor(runs[0][0]), 'losses dp vs single', tol=1e-6)
    assert_close(runs[1][1], runs[0][1], 'parameters dp vs single', tol=1e-6)


@pytest.mark.parametrize('transport', ['torch', 'rccl'])
def test_dp_eager_hooks(nccl_world1, transport, monkeypatch):
    """Eager mode: the engine's bucket hooks launch the all-reduces, wait() fences them."""
    monkeypatch.setenv('MVAE_COMM', transport)
    _, model, d = build_pair('mnist', weight_seed=43)
    eng = BimodalStep(model, 16, 1.0, 50.0, seed=3)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    dp = DataParallel(model, eng)
    image, label = OS.synthetic_batch('mnist', 16, seed=95)
    before = model.arena.flat.clone()
    eng.step(image.to(DEV), label.to(DEV), 0.5)
    assert len(dp.buckets.pending) == 2
    dp.finish(opt)
    assert not dp.buckets.pending and opt._step_dev.item() == 1
    torch.cuda.synchronize()
    assert not torch.equal(before, model.arena.flat)


# ----------------------------------------------------------------------------- two ranks, one GPU
LAM = {'mnist': 50.0, 'fashionmnist': 50.0, 'celeba': 10.0, 'celeba19': 10.0}
COMBO_SEED = 97531          # the SHARED subset seed of the celeba19 replicas


def _make_engine(kind, model, batch, rank):
    if kind == 'celeba19':
        from mvae_amd.engine import Celeba19Step
        return Celeba19Step(model, batch, 1.0, LAM[kind], approx_m=1, seed=5 + rank, combo_seed=COMBO_SEED)
    return BimodalStep(model, batch, 1.0, LAM[kind], seed=5 + rank)


def _shard_noise(kind, batch, d, rank, combos=None):
    torch.manual_seed(400 + rank)
    if kind == 'celeba19':
        return OS.draw_celeba19_noise(batch, d, OS.celeba19_terms(combos))
    return OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))


def _two_rank_worker(rank, world, port, kind, batch, out_dir):
    """One data-parallel replica.  Both ranks share cuda:0 (RCCL refuses two ranks per device, gloo
    stages CUDA tensors through the host), which is enough to run the REAL multi-rank code path --
    broadcast, bucket launches from the engine hooks, per-bucket fence + Adam, 1/N in FusedAdam -- on a
    one-GPU box."""
    import numpy as np
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # different initial weights per rank: the broadcast from rank 0 must make them equal
        oracle, model, d = build_pair(kind, weight_seed=70 + rank)
        eng = _make_engine(kind, model, batch, rank)
        opt = FusedAdam(model.parameters(), lr=1e-3, grad_scale=1.0 / world)
        dp = DataParallel(model, eng)
        image, label = OS.synthetic_batch(kind, batch, seed=300 + rank)       # this rank's shard
        if kind == 'celeba19':
            # the subsets of the step come from the engine's own generator: same seed on every rank
            from mvae_amd.engine import sample_subsets
            combos = sample_subsets(eng.rng, 19, 1)
            np.save(os.path.join(out_dir, 'combos_rank%d.npy' % rank), combos)
            eng.step(image.to(DEV), label.to(DEV), 0.5, noise=_shard_noise(kind, batch, d, rank, combos), combos=combos)
        else:
            eng.step(image.to(DEV), label.to(DEV), 0.5, noise=_shard_noise(kind, batch, d, rank))
        n_pending = len(dp.buckets.pending)
        dp.wait()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, 'grad_sum_rank%d.npy' % rank), model.arena.grad.cpu().numpy())
        np.save(os.path.join(out_dir, 'weights_rank%d.npy' % rank), model.arena.flat.detach().cpu().numpy())
        np.save(os.path.join(out_dir, 'buckets_rank%d.npy' % rank), np.asarray([dp.n_buckets, n_pending]))
        dp.finish(opt)          # nothing pending any more: Adam per bucket range + one counter advance
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, 'after_rank%d.npy' % rank), model.arena.flat.detach().cpu().numpy())
    finally:
        dist.destroy_process_group()


# the four experiments: BASELINE.json assigns fashionmnist (2 GPUs) and celeba19 (8 GPUs, 38 sub-modules,
# grouped launches) to multi-GPU runs first
@pytest.mark.parametrize('kind,batch,n_buckets', [('mnist', 16, 2), ('fashionmnist', 12, 3), ('celeba', 6, 3),
                                                  ('celeba19', 4, 3)])
def test_two_ranks_average_shard_gradients(kind, batch, n_buckets, tmp_path):
    """SURVEY section 8e parity definition: the N-replica gradient is the average over shards of the
    reference's per-shard gradients at shared weights, each shard with its own noise."""
    import numpy as np
    import torch.multiprocessing as mp
    from util import assert_zero_grad, is_zero_grad, zero_grad_weight
    world, port = 2, _free_port()
    mp.spawn(_two_rank_worker, args=(world, port, kind, batch, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(str(tmp_path / 'grad_sum_rank0.npy')); g1 = np.load(str(tmp_path / 'grad_sum_rank1.npy'))
    assert np.array_equal(g0, g1), 'ranks hold different reduced gradients'
    w0 = np.load(str(tmp_path / 'weights_rank0.npy')); w1 = np.load(str(tmp_path / 'weights_rank1.npy'))
    assert np.array_equal(w0, w1), 'broadcast did not equalise the replicas'
    a0 = np.load(str(tmp_path / 'after_rank0.npy')); a1 = np.load(str(tmp_path / 'after_rank1.npy'))
    assert np.array_equal(a0, a1) and not np.array_equal(a0, w0), 'replicas diverged after the optimizer step'
    for r in range(world):      # every bucket of the plan was launched exactly once by the engine's hooks
        assert np.load(str(tmp_path / ('buckets_rank%d.npy' % r))).tolist() == [n_buckets, n_buckets]
    combos = None
    if kind == 'celeba19':
        combos = np.load(str(tmp_path / 'combos_rank0.npy'))
        assert np.array_equal(combos, np.load(str(tmp_path / 'combos_rank1.npy'))), 'ranks drew different subsets'
    # oracle: rank 0's weights, both shards
    oracle, model, d = build_pair(kind, weight_seed=70)
    sums = None
    for rank in range(world):
        image, label = OS.synthetic_batch(kind, batch, seed=300 + rank)
        noise = _shard_noise(kind, batch, d, rank, combos)
        oracle.zero_grad()
        for m in oracle.modules():      # BatchNorm running statistics are per replica: restart them per shard
            if hasattr(m, 'reset_running_stats'):
                m.reset_running_stats()
        if kind == 'celeba19':
            total, _, _ = OS.celeba19_step(oracle, image, label, OS.celeba19_terms(combos), noise, 1.0, LAM[kind], 0.5)
        else:
            total, _, _ = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, LAM[kind], 0.5)
        total.backward()
        grads = {n: p.grad.clone() for n, p in oracle.named_parameters()}
        sums = grads if sums is None else {n: sums[n] + grads[n] for n in sums}
    model.finalize()
    flat = torch.from_numpy(g0)
    bad = []
    params = dict(model.named_parameters())

    def reduced(name):
        q = params[name]
        off = q.data_ptr() - model.arena.flat.data_ptr()
        return flat[off // 4:off // 4 + q.numel()].reshape(q.shape)

    for name, p in params.items():
        got = reduced(name)
        ref = sums[name]
        if is_zero_grad(kind, name):     # exactly zero in exact arithmetic: round-off on each side (tests/util.py)
            wn = zero_grad_weight(name)
            assert_zero_grad(name, got.abs().max().item(), reduced(wn).abs().max().item(), 'HIP, summed over ranks')
            assert_zero_grad(name, ref.abs().max().item(), sums[wn].abs().max().item(), 'oracle, summed over shards')
            continue
        scale = max(ref.abs().max().item(), 1e-30)
        err = (got - ref).abs().max().item() / scale
        if err > 1e-4:
            bad.append('%s %.3e' % (name, err))
    assert not bad, 'summed gradients beyond 1e-4: ' + '; '.join(bad)


def _two_rank_replay_worker(rank, world, port, kind, batch, out_dir):
    import numpy as np
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _, model, d = build_pair(kind, weight_seed=80 + rank)
        eng = _make_engine(kind, model, batch, rank)
        opt = FusedAdam(model.parameters(), lr=1e-3, grad_scale=1.0 / world)
        dp = DataParallel(model, eng)
        image, label = OS.synthetic_batch(kind, batch, seed=500 + rank)
        eng.capture(opt, image.shape[1:], label, comm=dp)          # what bench.py does for --gpus N > 1
        losses = []
        for step in range(4):
            image, label = OS.synthetic_batch(kind, batch, seed=510 + 10 * step + rank)
            losses.append(eng.replay(image.to(DEV), label.to(DEV), 0.25 * (step + 1))[-1].item())
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, 'replay_rank%d.npy' % rank), model.arena.flat.detach().cpu().numpy())
        np.save(os.path.join(out_dir, 'losses_rank%d.npy' % rank), np.asarray(losses))
        if kind == 'celeba19':
            np.save(os.path.join(out_dir, 'combos_rank%d.npy' % rank), eng.combos)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,batch', [('mnist', 32), ('fashionmnist', 8), ('celeba19', 4)])
def test_two_ranks_graph_replay_keeps_replicas_identical(kind, batch, tmp_path):
    """The launch structure of ``bench.py --gpus N``: one captured graph per gradient bucket with the
    all-reduces between them and Adam per bucket, two ranks with different shards and noise streams."""
    import numpy as np
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_two_rank_replay_worker, args=(world, port, kind, batch, str(tmp_path)), nprocs=world, join=True)
    p0 = np.load(str(tmp_path / 'replay_rank0.npy')); p1 = np.load(str(tmp_path / 'replay_rank1.npy'))
    assert np.array_equal(p0, p1), 'replicas diverged under graph replay'
    l0 = np.load(str(tmp_path / 'losses_rank0.npy')); l1 = np.load(str(tmp_path / 'losses_rank1.npy'))
    assert np.isfinite(l0).all() and np.isfinite(l1).all() and not np.array_equal(l0, l1)   # different shards
    if kind == 'celeba19':      # the shared subset seed: both ranks replayed the same terms in every step
        assert np.array_equal(np.load(str(tmp_path / 'combos_rank0.npy')), np.load(str(tmp_path / 'combos_rank1.npy')))

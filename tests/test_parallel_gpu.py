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
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize('kind,batch', [('mnist', 32), ('celeba', 8)])
def test_dp_three_graph_path_matches_single_graph(nccl_world1, kind, batch):
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
            assert len(eng._graphs) == 3
    assert_close(torch.tensor(runs[1][0]), torch.tensor(runs[0][0]), 'losses dp vs single', tol=1e-6)
    assert_close(runs[1][1], runs[0][1], 'parameters dp vs single', tol=1e-6)


def test_dp_eager_hooks(nccl_world1):
    """Eager mode: the engine's bucket hooks launch the all-reduces, wait() fences them."""
    _, model, d = build_pair('mnist', weight_seed=43)
    eng = BimodalStep(model, 16, 1.0, 50.0, seed=3)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    dp = DataParallel(model, eng)
    image, label = OS.synthetic_batch('mnist', 16, seed=95)
    before = model.arena.flat.clone()
    eng.step(image.to(DEV), label.to(DEV), 0.5)
    assert len(dp.buckets.pending) == 2
    dp.wait()
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(before, model.arena.flat)


# ----------------------------------------------------------------------------- two ranks, one GPU
def _two_rank_worker(rank, world, port, kind, batch, out_dir):
    """One data-parallel replica.  Both ranks share cuda:0 (RCCL refuses two ranks per device, gloo
    stages CUDA tensors through the host), which is enough to run the REAL multi-rank code path --
    broadcast, bucket launches from the engine hooks, wait, 1/N in FusedAdam -- on a one-GPU box."""
    import numpy as np
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lam = 50.0 if kind == 'mnist' else 10.0
        # different initial weights per rank: the broadcast from rank 0 must make them equal
        oracle, model, d = build_pair(kind, weight_seed=70 + rank)
        eng = BimodalStep(model, batch, 1.0, lam, seed=5 + rank)
        opt = FusedAdam(model.parameters(), lr=1e-3, grad_scale=1.0 / world)
        dp = DataParallel(model, eng)
        image, label = OS.synthetic_batch(kind, batch, seed=300 + rank)       # this rank's shard
        torch.manual_seed(400 + rank)
        noise = OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))
        eng.step(image.to(DEV), label.to(DEV), 0.5, noise=noise)
        dp.wait()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, 'grad_sum_rank%d.npy' % rank), model.arena.grad.cpu().numpy())
        np.save(os.path.join(out_dir, 'weights_rank%d.npy' % rank), model.arena.flat.detach().cpu().numpy())
        opt.step()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, 'after_rank%d.npy' % rank), model.arena.flat.detach().cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,batch', [('mnist', 16), ('celeba', 6)])
def test_two_ranks_average_shard_gradients(kind, batch, tmp_path):
    """SURVEY section 8e parity definition: the N-replica gradient is the average over shards of the
    reference's per-shard gradients at shared weights, each shard with its own noise."""
    import numpy as np
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_two_rank_worker, args=(world, port, kind, batch, str(tmp_path)), nprocs=world, join=True)
    g0 = np.load(str(tmp_path / 'grad_sum_rank0.npy')); g1 = np.load(str(tmp_path / 'grad_sum_rank1.npy'))
    assert np.array_equal(g0, g1), 'ranks hold different reduced gradients'
    w0 = np.load(str(tmp_path / 'weights_rank0.npy')); w1 = np.load(str(tmp_path / 'weights_rank1.npy'))
    assert np.array_equal(w0, w1), 'broadcast did not equalise the replicas'
    a0 = np.load(str(tmp_path / 'after_rank0.npy')); a1 = np.load(str(tmp_path / 'after_rank1.npy'))
    assert np.array_equal(a0, a1) and not np.array_equal(a0, w0), 'replicas diverged after the optimizer step'
    # oracle: rank 0's weights, both shards
    lam = 50.0 if kind == 'mnist' else 10.0
    oracle, model, d = build_pair(kind, weight_seed=70)
    sums = None
    for rank in range(world):
        image, label = OS.synthetic_batch(kind, batch, seed=300 + rank)
        torch.manual_seed(400 + rank)
        noise = OS.draw_bimodal_noise(batch, d, has_dropout=(kind == 'celeba'))
        oracle.zero_grad()
        if kind == 'celeba':       # BatchNorm running statistics are per replica: restart them per shard
            for m in oracle.modules():
                if hasattr(m, 'reset_running_stats'):
                    m.reset_running_stats()
        total, _, _ = OS.bimodal_step(oracle, kind, image, label, noise, 1.0, lam, 0.5)
        total.backward()
        grads = {n: p.grad.clone() for n, p in oracle.named_parameters()}
        sums = grads if sums is None else {n: sums[n] + grads[n] for n in sums}
    model.finalize()
    gmax = max(v.abs().max().item() for v in sums.values())
    flat = torch.from_numpy(g0)
    for name, p in model.named_parameters():
        off = p.data_ptr() - model.arena.flat.data_ptr()
        got = flat[off // 4:off // 4 + p.numel()].reshape(p.shape)
        ref = sums[name]
        scale = max(ref.abs().max().item(), 1e-2 * gmax)
        err = (got - ref).abs().max().item() / scale
        assert err <= 1e-4, 'summed gradient of %s: relative error %.3e' % (name, err)


def _two_rank_replay_worker(rank, world, port, kind, batch, out_dir):
    import numpy as np
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lam = 50.0 if kind == 'mnist' else 10.0
        _, model, d = build_pair(kind, weight_seed=80 + rank)
        eng = BimodalStep(model, batch, 1.0, lam, seed=9 + rank)
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
    finally:
        dist.destroy_process_group()


def test_two_ranks_graph_replay_keeps_replicas_identical(tmp_path):
    """The launch structure of ``bench.py --gpus N``: three captured graphs per step with the bucket
    all-reduces between them, two ranks with different shards and noise streams."""
    import numpy as np
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_two_rank_replay_worker, args=(world, port, 'mnist', 32, str(tmp_path)), nprocs=world, join=True)
    p0 = np.load(str(tmp_path / 'replay_rank0.npy')); p1 = np.load(str(tmp_path / 'replay_rank1.npy'))
    assert np.array_equal(p0, p1), 'replicas diverged under graph replay'
    l0 = np.load(str(tmp_path / 'losses_rank0.npy')); l1 = np.load(str(tmp_path / 'losses_rank1.npy'))
    assert np.isfinite(l0).all() and np.isfinite(l1).all() and not np.array_equal(l0, l1)   # different shards

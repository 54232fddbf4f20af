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

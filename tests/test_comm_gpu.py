"""GPU: the library's own RCCL communicator (``mvae_comm_*``, csrc/comm.hip -- SURVEY 8b) at world size 1 on the
one-GPU box: the C ABI end to end (unique id -> init -> broadcast -> all-reduce tickets -> wait -> destroy), the
same collectives captured inside a hipGraph, and the engine's ONE-graph data-parallel step (forward, backward,
bucket all-reduces on the communicator's stream, per-bucket Adam) against the single-GPU graph.  N > 1 needs N
GPUs (RCCL refuses two ranks on one device): the driver's multi-GPU bench is its hardware test; the rank logic is
covered by the gloo tests."""
import ctypes
import os
import socket

import pytest
import torch
import torch.distributed as dist

from mvae_amd import _lib
from mvae_amd.engine import BimodalStep
from mvae_amd.optim import FusedAdam
from mvae_amd.parallel import DataParallel, RcclBuckets, RcclComm
from oracle import steps as OS
from test_engine_gpu import build_pair
from util import assert_close, init_world1

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.fixture(scope='module')
def world1():
    init_world1('nccl', DEV)
    yield
    dist.destroy_process_group()


def test_c_abi_roundtrip_without_torch_distributed():
    """What a non-PyTorch host does: id, init, broadcast, all-reduce, wait, destroy -- torch only holds the memory."""
    L = _lib.lib()
    RcclComm.bind_torch_rccl()
    assert L.mvae_comm_rccl_version() > 20000
    ident = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
    assert L.mvae_comm_unique_id(ident, _lib.COMM_ID_BYTES) == 0
    assert L.mvae_comm_unique_id(ident, 64) == -1                      # buffer too small
    h = ctypes.c_void_p()
    assert L.mvae_comm_init(ctypes.byref(h), ident, _lib.COMM_ID_BYTES, 0, 1, 99) == -1      # no such device
    assert L.mvae_comm_init(ctypes.byref(h), ident, _lib.COMM_ID_BYTES, 1, 1, 0) == -1       # rank >= world
    assert L.mvae_comm_init(ctypes.byref(h), ident, _lib.COMM_ID_BYTES, 0, 1, 0) == 0
    assert L.mvae_comm_rank(h) == 0 and L.mvae_comm_world(h) == 1
    x = torch.arange(1000, dtype=torch.float32, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.mvae_comm_broadcast(h, ctypes.c_void_p(x.data_ptr()), x.numel() * 4, 0, st) == 0
    t1, t2 = ctypes.c_int(-1), ctypes.c_int(-1)
    assert L.mvae_comm_allreduce_async(h, ctypes.c_void_p(x.data_ptr()), 600, st, ctypes.byref(t1)) == 0
    assert L.mvae_comm_allreduce_async(h, ctypes.c_void_p(x.data_ptr() + 2400), 400, st, ctypes.byref(t2)) == 0
    assert (t1.value, t2.value) == (1, 2)                                # the broadcast took ticket 0
    assert L.mvae_comm_wait(h, t1.value, st) == 0 and L.mvae_comm_wait(h, -1, st) == 0
    assert L.mvae_comm_wait(h, 7, st) == -1                              # never issued
    assert L.mvae_comm_allreduce_async(h, None, 4, st, ctypes.byref(t1)) == -1
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32))   # sum over one rank
    assert L.mvae_comm_destroy(h) == 0


def test_comm_object_self_check_and_buckets(world1):
    comm = RcclComm.from_process_group(DEV)
    assert comm.world == 1 and comm.rccl_version.startswith('2.')
    comm.self_check(DEV)                                                   # eager + captured, known answer
    flat = torch.randn(1000, device=DEV)
    ref = flat.clone()
    b = RcclBuckets(flat, [(0, 300), (300, 1000)], comm)
    b.launch(0); b.launch(1)
    with pytest.raises(RuntimeError, match='twice'):
        b.launch(1)
    b.wait(0); b.wait()
    torch.cuda.synchronize()
    assert torch.equal(flat, ref) and not b.pending
    with pytest.raises(ValueError):
        RcclBuckets(flat, [(0, 300), (400, 1000)], comm)
    comm.destroy()


@pytest.mark.parametrize('kind,batch', [('mnist', 32), ('celeba', 8)])
def test_one_graph_data_parallel_step_equals_single_gpu_graph(world1, kind, batch, monkeypatch):
    """mvae_comm transport: the whole data-parallel step is ONE captured graph; at world size 1 it must train
    exactly like the single-GPU graph (same Philox stream, same kernels; Adam per bucket range is the same
    elementwise update).  The torch.distributed three-graph transport stays available (MVAE_COMM=torch)."""
    lam = 50.0 if kind == 'mnist' else 10.0
    runs = {}
    for mode in ('single', 'rccl', 'torch'):
        monkeypatch.setenv('MVAE_COMM', 'torch' if mode == 'torch' else 'rccl')
        _, model, d = build_pair(kind, weight_seed=41)
        eng = BimodalStep(model, batch, 1.0, lam, seed=7)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        dp = DataParallel(model, eng) if mode != 'single' else None
        image, label = OS.synthetic_batch(kind, batch, seed=90)
        eng.capture(opt, image.shape[1:], label, comm=dp)
        losses = []
        for step in range(3):
            image, label = OS.synthetic_batch(kind, batch, seed=91 + step)
            losses.append(eng.replay(image.to(DEV), label.to(DEV), 0.25 * (step + 1))[-1].item())
        torch.cuda.synchronize()
        runs[mode] = (losses, model.arena.flat.clone(), opt._step_dev.item())
        if mode == 'rccl':
            assert dp.in_graph and len(eng._graphs) == 1 and 'mvae_comm' in dp.transport
        if mode == 'torch':
            assert not dp.in_graph and len(eng._graphs) == dp.n_buckets
    assert runs['rccl'][2] == runs['single'][2] == 3
    assert_close(torch.tensor(runs['rccl'][0]), torch.tensor(runs['single'][0]), 'losses one-graph dp vs single', tol=1e-6)
    assert_close(runs['rccl'][1], runs['single'][1], 'parameters one-graph dp vs single', tol=1e-6)
    assert torch.equal(runs['rccl'][1], runs['torch'][1]), 'the two transports must give the same bits at world 1'


def test_rank_supervisor_on_the_gpu_kills_a_hung_attempt_and_the_next_one_gets_the_gpu():
    """The transport chain of ``bench.py --gpus N`` (mvae_amd/launch.py) end to end on a GPU, at the only world size a
    one-GPU box offers: under the real launcher the rank is a supervisor; its first child hangs (after it has a GPU context
    and an RCCL process group) and is killed with its process group when the budget is spent; the second child -- the
    library's communicator, collectives inside the step graph -- must find the GPU usable and deliver the line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'MVAE_COMM'):
        env.pop(k, None)
    env.update(MVAE_BENCH_SUPERVISE='force', MVAE_BENCH_CHAIN='fake-hang:25,mvae_comm-one-graph:200')
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--force-dp', '--no-extras',
           '--steps', '10', '--warmup', '3']
    r = subprocess.run(cmd, env=env, timeout=400, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    line = json.loads(lines[-1])
    assert line['value'] and line['n_gpus'] == 1
    d = line['dist']
    assert d['transport_attempt'] == 'mvae_comm-one-graph' and 'mvae_comm' in d['transport']
    assert [t['transport'] for t in d['fallbacks_tried']] == ['fake-hang'] and d['fallbacks_tried'][0]['ranks'] == ['timeout']

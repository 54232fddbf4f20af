"""CPU: the N > 1 path with world_size 2 over gloo -- bucketed asynchronous gradient all-reduce
on a flat arena, the averaged-gradient definition of SURVEY.md section 8(e), and the launch
order the engine uses (bucket 0 = decoders before bucket 1 = encoders)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import mvae_amd
    from mvae_amd.arena import ParamArena
    from mvae_amd.parallel import GradBuckets, bucket_ranges

    torch.manual_seed(100 + rank)                      # replicas start from DIFFERENT weights ...
    model = mvae_amd.mnist.model.MVAE(16)
    arena = ParamArena(model, order=model.arena_order(), adjacent=model.arena_adjacent())
    dist.broadcast(arena.flat, src=0)                  # ... and are made identical, as DataParallel does
    ranges = bucket_ranges(model, arena)
    buckets = GradBuckets(arena.grad, ranges)

    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(arena.numel, generator=g)
    arena.grad.copy_(local)
    buckets.launch(0)                                  # decoders' range first (ready first in backward)
    buckets.launch(1)
    buckets.wait()
    averaged = arena.grad * (1.0 / world)              # FusedAdam's grad_scale
    torch.save({'flat': arena.flat.clone(), 'avg': averaged, 'local': local, 'ranges': ranges},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % i)) for i in range(world)]
    assert torch.equal(r[0]['flat'], r[1]['flat']), 'parameters must be identical after the broadcast'
    expect = (r[0]['local'] + r[1]['local']) / world
    for i in range(world):
        assert torch.allclose(r[i]['avg'], expect, rtol=0, atol=1e-6)
    (a0, a1), (b0, b1) = r[0]['ranges']
    assert a0 == 0 and a1 == b0 and b1 == r[0]['flat'].numel()


def test_bucket_ranges_must_tile_the_arena():
    from mvae_amd.parallel import GradBuckets
    flat = torch.zeros(100)
    GradBuckets(flat, [(0, 40), (40, 100)])
    with pytest.raises(ValueError):
        GradBuckets(flat, [(0, 40), (50, 100)])
    with pytest.raises(ValueError):
        GradBuckets(flat, [(0, 40), (40, 90)])


@pytest.mark.parametrize('kind,n_latents,n_buckets', [('mnist', 64, 2), ('fashionmnist', 64, 3), ('celeba', 100, 3),
                                                      ('celeba19', 100, 3)])
def test_bucket_plan_per_model(kind, n_latents, n_buckets):
    """Buckets follow backward completion: decoders | encoders | the image encoder's first layers (arena tail,
    laid out LAST so the final all-reduce is the smallest); MNIST's 4 MB of encoders stay one bucket."""
    sys.path.insert(0, ROOT)
    import mvae_amd
    from mvae_amd.arena import ParamArena
    from mvae_amd.parallel import GradBuckets, bucket_ranges
    model = getattr(mvae_amd, kind).model.MVAE(n_latents)
    arena = ParamArena(model, order=model.arena_order(), adjacent=model.arena_adjacent(), tail=model.arena_tail())
    ranges = bucket_ranges(model, arena)
    assert len(ranges) == n_buckets
    GradBuckets(arena.grad, ranges)                     # tiles the arena, no gaps
    tail_params = [p for m in model.arena_tail() for p in m.parameters()]
    assert tail_params and arena.tail_range[1] == arena.numel
    for p in tail_params:
        assert arena.tail_range[0] <= p._arena_off and p._arena_off + p.numel() <= arena.numel
    if n_buckets == 3:
        assert ranges[2] == arena.tail_range
        sizes = [hi - lo for lo, hi in ranges]
        assert sizes[2] < sizes[1] and sizes[2] * 4 < (4 << 20)      # the exposed collective is a few MB at most
    # every decoder parameter is in bucket 0, no encoder parameter is
    for name, p in model.named_parameters():
        in0 = ranges[0][0] <= p._arena_off < ranges[0][1]
        assert in0 == ('decoder' in name), name


def _finish_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mvae_amd.parallel import GradBuckets
    flat = torch.full((90,), float(rank + 1))
    b = GradBuckets(flat, [(0, 40), (40, 80), (80, 90)])
    order = []
    for k in range(3):
        b.launch(k)
    for k in range(3):                                  # per-bucket fences, in bucket order (DataParallel.finish)
        b.wait(k)
        order.append(float(flat[b.ranges[k][0]]))
    assert not b.pending
    with pytest.raises(RuntimeError):
        b.launch(0); b.launch(0)
    b.wait()
    torch.save(order, os.path.join(out_dir, 'order%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_per_bucket_fences_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_finish_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), 'order%d.pt' % r)) == [3.0, 3.0, 3.0]


# ----------------------------------------------------------------------------- bench.py as its own launcher
def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + argv, env=env, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_n_spawns_n_ranks_by_itself():
    """``python bench.py --gpus 2`` with no launcher (it silently ran ONE rank): the script re-runs
    itself as two ranks under torch.distributed.run; the gloo backend stands in for RCCL on this CPU-only box
    and the line shows two ranks that really exchanged data."""
    rc, line, err = _run_bench(['--gpus', '2', '--backend', 'gloo'])
    assert rc == 0, err[-2000:]
    assert line['n_gpus'] == 2
    assert line['dist']['world_size'] == 2 and line['dist']['allreduce_of_ones'] == 2.0


def test_bench_refuses_more_gpus_than_visible():
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-box check')
    rc, line, err = _run_bench(['--gpus', '2'])
    assert rc != 0 and line is None and 'only 0 GPU(s) visible' in err


def test_bench_refuses_a_launcher_that_disagrees_with_the_flag():
    rc, line, err = _run_bench(['--gpus', '4', '--backend', 'gloo'],
                               {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert rc != 0 and line is None and 'WORLD_SIZE=1' in err


# ----------------------------------------------------------------------------- the time-boxed transport chain
def test_transport_chain_survives_a_hang_and_a_crash():
    """the first N > 1 run must not be able to return nothing.  Every launched rank is a supervisor
    (mvae_amd/launch.py) that runs each transport in a child under a wall-clock budget.  Here the first transport
    HANGS (one rank never reaches the collective, its peer blocks in it), the second RAISES, the third works: the line
    comes from the third, within the budgets, and says what was given up on."""
    import time
    t0 = time.time()
    rc, line, err = _run_bench(['--gpus', '2', '--backend', 'gloo'],
                               {'MVAE_BENCH_CHAIN': 'fake-hang:8,fake-raise:60,default:120'}, timeout=300)
    wall = time.time() - t0
    assert rc == 0, err[-3000:]
    assert line['n_gpus'] == 2 and line['dist']['allreduce_of_ones'] == 2.0
    tried = line['dist']['fallbacks_tried']
    # (an attempt that dies within seconds with rc != 0 is granted ONE retry on a fresh rendezvous port: launch.py)
    assert [t['transport'] for t in tried if not t.get('retry')] == ['fake-hang', 'fake-raise']
    assert [t['transport'] for t in tried if t.get('retry')] in ([], ['fake-raise'])
    assert 'timeout' in tried[0]['ranks'] or 'peer-failed' in tried[0]['ranks']
    assert any(v.startswith('rc=') for v in tried[1]['ranks'])
    assert line['dist']['transport_attempt'] == 'default'
    assert tried[0]['wall_s'] < 8 + 30 and wall < 200, (tried, wall)


def test_transport_chain_reports_when_every_transport_fails():
    rc, line, err = _run_bench(['--gpus', '2', '--backend', 'gloo'],
                               {'MVAE_BENCH_CHAIN': 'fake-raise:60,fake-hang:6'}, timeout=300)
    assert line is not None and line['value'] is None and line['n_gpus'] == 2
    assert [t['transport'] for t in line['dist']['fallbacks_tried'] if not t.get('retry')] == ['fake-raise', 'fake-hang']


def test_run_attempt_kills_the_whole_process_group_on_timeout(tmp_path):
    """A child that spawns a grandchild and hangs: both are gone after the budget (the supervisor made the process
    group itself, so killing it cannot touch anything else)."""
    import sys
    import time
    import mvae_amd  # noqa: F401
    from mvae_amd import launch
    pidfile = tmp_path / 'pids'
    code = ('import os, subprocess, sys, time\n'
            'p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])\n'
            'open(%r, "w").write("%%d %%d" %% (os.getpid(), p.pid))\n'
            'print("{\\"partial\\": 1}"); sys.stdout.flush()\n'
            'time.sleep(600)\n' % str(pidfile))
    t0 = time.time()
    status, out = launch.run_attempt([sys.executable, '-c', code], dict(os.environ), 3.0)
    assert status == 'timeout' and time.time() - t0 < 30
    assert launch.last_json_line(out) == {'partial': 1}
    for pid in (int(v) for v in pidfile.read_text().split()):
        for _ in range(50):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            time.sleep(0.1)
        else:
            # a zombie still answers kill(0); it must at least not be running
            assert open('/proc/%d/stat' % pid).read().split()[2] in ('Z', 'X'), 'pid %d survived' % pid
    status, out = launch.run_attempt([sys.executable, '-c', 'print("{\\"a\\": 2}")'], dict(os.environ), 30.0)
    assert status == 'ok' and launch.last_json_line(out) == {'a': 2}
    status, _ = launch.run_attempt([sys.executable, '-c', 'raise SystemExit(7)'], dict(os.environ), 30.0)
    assert status == 'rc=7'
    assert [a.name for a in launch.parse_chain('fake-hang:2, default')] == ['fake-hang', 'default']
    with pytest.raises(ValueError):
        launch.parse_chain('no-such-transport')


# ----------------------------------------------------------------------------- failure-safe communicator rendezvous
class _FakeCommLib(object):
    """Stands in for libmvae_hip.so's mvae_comm_* entry points: which step fails on which rank is the test's choice."""
    def __init__(self, rank, scenario, log):
        self.rank, self.scenario, self.log = rank, scenario, log

    def mvae_comm_use_library(self, path):
        return 0

    def mvae_comm_rccl_version(self):
        return -5 if (self.scenario == 'bind-fails-on-1' and self.rank == 1) else 22105

    def mvae_comm_unique_id(self, buf, n):
        if self.scenario == 'id-fails-on-0':
            return -5
        for i in range(n):
            buf[i] = (i * 7 + 3) % 251
        return 0

    def mvae_comm_init(self, h, uid, n, rank, world, device):
        self.log.append('init')
        assert bytes(uid)[:4] == bytes([(i * 7 + 3) % 251 for i in range(4)])
        return -5 if (self.scenario == 'init-fails-on-1' and self.rank == 1) else 0

    def mvae_comm_destroy(self, h):
        self.log.append('destroy')
        return 0

    def mvae_comm_last_error(self, h):
        return b'fake'


def _rendezvous_worker(rank, world, port, scenario, outdir):
    import torch
    import torch.distributed as dist
    import mvae_amd  # noqa: F401
    from mvae_amd import _lib, parallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    log = []
    fake = _FakeCommLib(rank, scenario, log)
    _lib.lib = lambda: fake
    try:
        comm = parallel.RcclComm.from_process_group(torch.device('cpu'))
        outcome = 'comm rank %d/%d' % (comm.rank, comm.world)
    except RuntimeError as e:
        outcome = 'raised: %s' % e
    # whatever happened, the ranks are still in step: one more collective completes
    t = torch.ones(1)
    dist.all_reduce(t)
    with open(os.path.join(outdir, 'r%d' % rank), 'w') as f:
        f.write('%s | %s | %g' % (outcome, ','.join(log), t.item()))
    dist.destroy_process_group()


@pytest.mark.parametrize('scenario,expect', [
    ('ok', ('comm rank 0/2', 'comm rank 1/2')),
    ('bind-fails-on-1', ('raised: binding RCCL / creating the unique id failed on a peer',
                         'raised: binding RCCL / creating the unique id failed on this rank')),
    ('id-fails-on-0', ('raised: binding RCCL / creating the unique id failed on this rank',
                       'raised: binding RCCL / creating the unique id failed on this rank')),
    ('init-fails-on-1', ('raised: mvae_comm_init failed on a peer', 'raised: mvae_comm_init failed on this rank'))])
def test_communicator_rendezvous_is_collective_safe(scenario, expect, tmp_path):
    """a rank that cannot bind RCCL / make the id / init must not leave its peers in a different
    collective.  All ranks raise together (or none), nobody enters ncclCommInitRank after a failed vote, and the process
    group is still usable afterwards."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_rendezvous_worker, args=(world, port, scenario, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        outcome, log, total = open(os.path.join(str(tmp_path), 'r%d' % r)).read().split(' | ')
        assert outcome.startswith(expect[r]), (r, outcome)
        assert float(total) == 2.0
        if scenario in ('bind-fails-on-1', 'id-fails-on-0'):
            assert 'init' not in log          # nobody entered the collective init
        if scenario == 'init-fails-on-1' and r == 0:
            assert log == 'init,destroy'      # the healthy rank gave its communicator back

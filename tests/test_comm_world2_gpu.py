"""GPU: the DEFAULT data-parallel transport -- the library's communicator (``mvae_comm_*``, csrc/comm.hip) with the
bucket all-reduces captured INSIDE the step's ONE hipGraph and Adam per bucket -- at WORLD SIZE 2 ("What's
missing" 1: that code path had only ever run with one rank; the world-2 tests of test_parallel_gpu.py go through gloo,
i.e. the three-graph torch.distributed transport).

RCCL refuses two ranks on one device and the box has one GPU, so the communicator is pointed (``MVAE_RCCL_LIB``, the
same switch a host uses to choose its librccl) at ``tests/shm_nccl/libshm_nccl.so``: the NCCL ABI subset comm.hip
binds, implemented over POSIX shared memory + host staging (test infrastructure; sums in rank order).  Everything
above that -- unique id over torch.distributed, ``mvae_comm_init``, the broadcast, tickets / events, the capture of the
collectives as graph nodes on the communicator's stream, per-bucket Adam with 1/N folded in, the watchdog -- is the
product code, two processes on cuda:0.

Parity definition: the reduced gradient = the SUM over shards of the oracle's per-shard gradients at the
shared (rank-0) weights, each shard with the noise its replica drew; 1e-4 relative per parameter.  Replicas stay
bit-identical over replays.  A rank that dies between steps makes its peer RAISE within the watchdog budget."""
import os
import socket
import time

import numpy as np
import pytest
import torch

from util import assert_zero_grad, is_zero_grad, note_redraws, zero_grad_weight

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LR = 1e-3
LAM = {'mnist': 50.0, 'fashionmnist': 50.0, 'celeba': 10.0, 'celeba19': 10.0}
BETA = 0.5
STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shm_nccl', 'libshm_nccl.so')
COMBO_SEED = 24680


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _engine(kind, model, batch, rank):
    from mvae_amd.engine import BimodalStep, Celeba19Step
    if kind == 'celeba19':
        return Celeba19Step(model, batch, 1.0, LAM[kind], approx_m=1, seed=31 + rank, combo_seed=COMBO_SEED)
    return BimodalStep(model, batch, 1.0, LAM[kind], seed=31 + rank)


def _worker(rank, world, port, kind, batch, input_seed, out_dir, die_after):
    """One replica.  ``die_after`` >= 0: rank 1 leaves (os._exit) after that many replays, rank 0 keeps stepping."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['MVAE_RCCL_LIB'] = STUB              # what mvae_comm_* dlopen-s instead of librccl
    os.environ['MVAE_SHMNCCL_TIMEOUT_S'] = '6' if die_after >= 0 else '60'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mvae_amd.optim import FusedAdam
    from mvae_amd.parallel import DataParallel
    from oracle import steps as OS
    from test_engine_gpu import build_pair, hits_bce_jump
    from test_replay_parity_gpu import _noise_in_reference_order
    _, model, d = build_pair(kind, weight_seed=80 + rank)          # the broadcast must make the replicas equal
    eng = _engine(kind, model, batch, rank)
    opt = FusedAdam(model.parameters(), lr=LR, grad_scale=1.0 / world)
    dp = DataParallel(model, eng, transport='rccl')
    assert dp.comm is not None and dp.in_graph, dp.transport
    assert 'mvae_comm' in dp.transport and dp.comm.rccl_version == '1.0.0', dp.transport      # the stand-in, not an RCCL
    w0 = model.arena.flat.detach().clone()
    image, label = OS.synthetic_batch(kind, batch, seed=input_seed + rank)        # this rank's shard
    eng.capture(opt, image.shape[1:], label, comm=dp)
    assert len(eng._graphs) == 1, 'the data-parallel step must be ONE graph with the collectives inside'
    assert torch.equal(w0, model.arena.flat), 'capture() left a trace in the parameters'
    t_raise = None
    for step in range(3):
        if die_after >= 0 and rank == 1 and step == die_after:
            os._exit(0)                                 # killed between two steps: no goodbye to the peer
        if step > 0:
            image, label = OS.synthetic_batch(kind, batch, seed=input_seed + 100 * step + rank)
        t0 = time.monotonic()
        try:
            elbo = eng.replay(image.to(DEV), label.to(DEV), BETA)
            dp.synchronize(timeout_s=60.0 if die_after < 0 else 30.0)       # the watchdog instead of torch.cuda.synchronize()
        except RuntimeError as e:
            t_raise = (step, time.monotonic() - t0, str(e))
            break
        if step == 0:
            combos = eng.combos if kind == 'celeba19' else None
            noise, _ = _noise_in_reference_order(kind, eng, combos)
            if kind != 'celeba19':
                elbo = eng.terms_in_reference_order(elbo)
            torch.save({'noise': noise, 'combos': combos, 'elbo': elbo.detach().cpu().clone(),
                        'grad': model.arena.grad.detach().cpu().clone(), 'w0': w0.cpu(),
                        'w1': model.arena.flat.detach().cpu().clone(),
                        'jump': bool(hits_bce_jump(eng)) if kind != 'celeba19' else False,
                        'ranges': list(dp.buckets.ranges), 'step_dev': int(opt._step_dev.item())},
                       os.path.join(out_dir, 'step0_rank%d.pt' % rank))
    if die_after >= 0:
        torch.save({'raised': t_raise}, os.path.join(out_dir, 'watchdog_rank%d.pt' % rank))
        dp.comm.abandon()
        os._exit(0)                                     # the gloo group has lost its peer too: nothing to tear down with
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, 'after_rank%d.npy' % rank), model.arena.flat.detach().cpu().numpy())
    dp.comm.destroy()
    dist.destroy_process_group()


def _spawn(kind, batch, input_seed, out_dir, die_after=-1, timeout=600):
    import subprocess
    import torch.multiprocessing as mp
    if not os.path.exists(STUB):        # normally built by __graft_entry__.build(); the box has the same toolchain
        subprocess.run(['make', '-C', os.path.dirname(STUB)], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert os.path.exists(STUB), '%s is not built: run __graft_entry__.build()' % STUB
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, batch, input_seed, str(out_dir), die_after)) for r in range(2)]
    for p in procs:
        p.start()
    deadline = time.monotonic() + timeout
    for p in procs:
        p.join(max(1.0, deadline - time.monotonic()))
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, 'a rank hung (killed after %d s)' % timeout
    assert [p.exitcode for p in procs] == [0, 0], 'rank exit codes %r' % [p.exitcode for p in procs]


def _oracle_shard_sums(kind, batch, input_seed, shards):
    """Sum over the two shards of the oracle's gradients at rank 0's weights, each shard on the noise its replica drew."""
    from oracle import steps as OS
    from test_engine_gpu import build_pair
    oracle, model, d = build_pair(kind, weight_seed=80)
    sums, zero_logit = None, False
    for rank, sh in enumerate(shards):
        image, label = OS.synthetic_batch(kind, batch, seed=input_seed + rank)
        oracle.zero_grad()
        for m in oracle.modules():          # BatchNorm running statistics are per replica
            if hasattr(m, 'reset_running_stats'):
                m.reset_running_stats()
        if kind == 'celeba19':
            terms = OS.celeba19_terms(sh['combos'])
            total, elbos, _ = OS.celeba19_step(oracle, image, label, terms, sh['noise'], 1.0, LAM[kind], BETA)
            got = sh['elbo']
            T = len(terms)
        else:
            total, elbos, _, recon = OS.bimodal_step(oracle, kind, image, label, sh['noise'], 1.0, LAM[kind], BETA,
                                                     return_recon=True)
            T = 3
            logits = [r[0] for r in recon if r[0] is not None]
            if kind == 'celeba':
                logits += [r[1] for r in recon if r[1] is not None]
            zero_logit = zero_logit or any(bool((x == 0).any()) for x in logits)
            got = sh['elbo']
        total.backward()
        ref_total = total.detach()
        hip_total = sh['elbo'][-1]
        assert abs(hip_total.item() - ref_total.item()) <= 1e-4 * abs(ref_total.item()), \
            'rank %d total ELBO %r vs oracle %r' % (rank, hip_total.item(), ref_total.item())
        if got is not None:
            ref = torch.stack(elbos).detach()
            assert (got[:T] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), 'rank %d ELBO terms' % rank
        grads = {n: p.grad.clone() for n, p in oracle.named_parameters()}
        sums = grads if sums is None else {n: sums[n] + grads[n] for n in sums}
    return model, sums, zero_logit


# (the last three: BASELINE.json configs[1..3]'s per-GPU batches on the default one-graph transport -- VERDICT r5 item 8b)
@pytest.mark.parametrize('kind,batch', [('mnist', 16), ('fashionmnist', 12), ('celeba', 6), ('celeba19', 4), ('mnist', 512),
                                        ('fashionmnist', 1024), ('celeba', 256)])
def test_one_graph_step_at_world_size_two(kind, batch, tmp_path):
    redraws = 0
    for attempt in range(3):
        out = tmp_path / ('try%d' % attempt)
        out.mkdir()
        input_seed = 700 + 7 * attempt
        _spawn(kind, batch, input_seed, out)
        shards = [torch.load(str(out / ('step0_rank%d.pt' % r)), weights_only=False) for r in range(2)]
        model, sums, zero_logit = _oracle_shard_sums(kind, batch, input_seed, shards)
        # an exactly-zero logit on either side sits on the reference BCE's gradient jump: re-draw, count
        if not (zero_logit or any(sh['jump'] for sh in shards)):
            break
        redraws += 1
    else:
        pytest.fail('three consecutive draws with an exactly-zero logit')
    note_redraws('world-2 one-graph %s B=%d' % (kind, batch), redraws)
    assert redraws <= 1
    a, b = shards
    assert a['step_dev'] == b['step_dev'] == 1
    assert torch.equal(a['w0'], b['w0']), 'broadcast did not equalise the replicas'
    assert torch.equal(a['grad'], b['grad']), 'ranks hold different reduced gradients'
    assert torch.equal(a['w1'], b['w1']) and not torch.equal(a['w1'], a['w0']), 'replicas diverged in the first step'
    assert a['ranges'] == b['ranges'] and len(a['ranges']) >= 2
    # ---- the reduced gradient = sum over shards of the oracle's per-shard gradients (1e-4)
    model.finalize()
    flat = a['grad']
    params = dict(model.named_parameters())

    def reduced(name):
        q = params[name]
        off = (q.data_ptr() - model.arena.flat.data_ptr()) // 4
        return flat[off:off + q.numel()].reshape(q.shape)

    bad, worst = [], 0.0
    for name in params:
        got, ref = reduced(name), sums[name]
        if is_zero_grad(kind, name):
            wn = zero_grad_weight(name)
            assert_zero_grad(name, got.abs().max().item(), reduced(wn).abs().max().item(), 'HIP, summed over ranks')
            assert_zero_grad(name, ref.abs().max().item(), sums[wn].abs().max().item(), 'oracle, summed over shards')
            continue
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        worst = max(worst, err)
        if err > 1e-4:
            bad.append('%s %.3e' % (name, err))
    assert not bad, 'reduced gradients beyond 1e-4: ' + '; '.join(bad)
    # ---- per-bucket Adam inside the graph, 1/N folded into its gradient read: torch.optim.Adam on the MEAN gradient
    p = torch.nn.Parameter(a['w0'].clone())
    p.grad = a['grad'] / 2.0
    torch.optim.Adam([p], lr=LR).step()
    bound = 1e-5 * LR + 2.4e-7 * p.detach().abs()
    excess = ((a['w1'] - p.detach()).abs() - bound).max().item()
    assert excess <= 0, 'in-graph Adam vs torch.optim.Adam on the mean gradient: %.3e beyond 2 ulp' % excess
    # ---- three replays: the replicas are still the same bits
    p0 = np.load(str(out / 'after_rank0.npy')); p1 = np.load(str(out / 'after_rank1.npy'))
    assert np.array_equal(p0, p1), 'replicas diverged under graph replay'
    assert not np.array_equal(p0, a['w1'].numpy())
    print('%s B=%d world 2, ONE graph with the collectives inside: worst reduced-gradient rel err %.2e, %d re-draw(s)'
          % (kind, batch, worst, redraws))


def test_reduce_scatter_all_gather_algorithm_gives_the_same_bits(tmp_path, monkeypatch):
    """MVAE_COMM_ALGO=rs_ag (csrc/comm.hip: the bucket all-reduce as an in-place ncclReduceScatter + ncclAllGather, the
    remainder through ncclAllReduce) -- the switch SURVEY 8e's "direct over ring" can be A/B-ed with the day a multi-GPU
    node exists.  Over the stand-in both algorithms add in rank order, so the reduced gradient, the first update and the
    weights after three replays must be the SAME BITS as the default's (FashionMNIST: three buckets, one of them with a
    length that is not a multiple of the world size)."""
    kind, batch, seed = 'fashionmnist', 12, 733
    outs = {}
    for algo in ('default', 'rs_ag'):
        out = tmp_path / algo
        out.mkdir()
        if algo == 'rs_ag':
            monkeypatch.setenv('MVAE_COMM_ALGO', 'rs_ag')
        else:
            monkeypatch.delenv('MVAE_COMM_ALGO', raising=False)
        _spawn(kind, batch, seed, out)
        outs[algo] = ([torch.load(str(out / ('step0_rank%d.pt' % r)), weights_only=False) for r in range(2)],
                      np.load(str(out / 'after_rank0.npy')))
    (d_sh, d_after), (r_sh, r_after) = outs['default'], outs['rs_ag']
    assert torch.equal(r_sh[0]['grad'], r_sh[1]['grad']) and torch.equal(r_sh[0]['w1'], r_sh[1]['w1'])
    assert torch.equal(d_sh[0]['grad'], r_sh[0]['grad']), 'rs_ag reduced a different gradient'
    assert torch.equal(d_sh[0]['w1'], r_sh[0]['w1']) and np.array_equal(d_after, r_after)


def test_peer_killed_between_steps_raises_within_the_watchdog_budget(tmp_path):
    """Rank 1 is gone after the first replay.  Rank 0's next step reaches a collective its peer will never join: the
    watchdog (``DataParallel.synchronize`` -> ``mvae_comm_synchronize``) must raise -- not hang, not return garbage
    silently -- within its budget (30 s here; the stand-in library gives up on the peer after 6 s)."""
    _spawn('mnist', 16, 900, tmp_path, die_after=1, timeout=240)
    res = torch.load(str(tmp_path / 'watchdog_rank0.pt'), weights_only=False)['raised']
    assert res is not None, 'rank 0 finished three steps although its peer had left after the first'
    step, seconds, text = res
    assert step == 1, res
    assert seconds < 30.0 + 5.0, 'raised after %.1f s' % seconds
    assert 'mvae_comm_synchronize' in text and ('peer' in text or 'watchdog' in text or 'remote' in text.lower()), text
    assert os.path.exists(str(tmp_path / 'step0_rank1.pt')) and not os.path.exists(str(tmp_path / 'watchdog_rank1.pt'))

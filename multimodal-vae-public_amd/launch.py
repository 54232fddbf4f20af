"""Time-boxed transport fallback chain for multi-rank runs (host logic only: no GPU, no RCCL call in here).

The data-parallel step has three ways to exchange gradients (parallel.py): the library's RCCL communicator with the
collectives inside ONE step graph, ``torch.distributed`` all-reduces between three graphs, and eager launches.  A
collective that HANGS (a peer that never arrives, a captured RCCL kernel that spins) cannot be recovered from inside
the process that issued it -- so a rank started by the launcher does no GPU work itself: it is a SUPERVISOR that
runs each transport in a child process under a wall-clock budget, kills the child's process group when the budget
is spent, and agrees with its peers -- through the launcher's own rendezvous store (env://) -- whether the
attempt succeeded on EVERY rank before anyone moves on.  The first attempt all ranks finished wins; its JSON line is
printed by rank 0 with ``dist.transport`` and ``dist.fallbacks_tried`` filled in.

    attempts = default_chain()                    # or parse_chain(os.environ['MVAE_BENCH_CHAIN'])
    line, tried = supervise(argv, attempts)       # called on every launched rank

Children find ``MVAE_BENCH_WORKER=1``, the attempt's name in ``MVAE_BENCH_TRANSPORT`` and a rendezvous port of
their own in ``MASTER_PORT`` (a port per attempt: a killed attempt's listener may linger).  The chain is exercised on
CPU with the fake transports ``fake-hang`` / ``fake-raise`` (tests/test_parallel_cpu.py).
"""
import datetime
import json
import os
import signal
import socket
import subprocess
import sys
import time

WORKER_ENV = 'MVAE_BENCH_WORKER'
TRANSPORT_ENV = 'MVAE_BENCH_TRANSPORT'
CHAIN_ENV = 'MVAE_BENCH_CHAIN'

# name -> (environment of the child, extra argv, default budget in seconds)
TRANSPORTS = {
    # budgets: a healthy 8-rank run (RCCL bring-up, capture, warm-up, timed steps, the collectives-off re-measure) takes
    # well under a minute once the supervisor's own `import torch` has paged the libraries in
    'mvae_comm-one-graph': ({'MVAE_COMM': 'rccl'}, [], 180.0),
    'torch-three-graphs': ({'MVAE_COMM': 'torch'}, [], 150.0),
    'torch-eager': ({'MVAE_COMM': 'torch'}, ['--no-graph'], 150.0),
    # CPU tests of the chain itself
    'fake-hang': ({}, [], 5.0),
    'fake-raise': ({}, [], 60.0),
    'default': ({}, [], 240.0),
}


class Attempt(object):
    def __init__(self, name, budget_s=None):
        if name not in TRANSPORTS:
            raise ValueError('unknown transport %r (known: %s)' % (name, ', '.join(sorted(TRANSPORTS))))
        self.name = name
        self.env, self.argv, default_budget = TRANSPORTS[name]
        self.budget_s = float(default_budget if budget_s is None else budget_s)


def default_chain():
    return [Attempt('mvae_comm-one-graph'), Attempt('torch-three-graphs'), Attempt('torch-eager')]


def parse_chain(spec):
    """``"name[:budget_s],name[:budget_s],..."`` -> [Attempt]."""
    out = []
    for part in spec.split(','):
        part = part.strip()
        if not part:
            continue
        name, _, budget = part.partition(':')
        out.append(Attempt(name, float(budget) if budget else None))
    if not out:
        raise ValueError('empty transport chain')
    return out


def chain_from_env():
    spec = os.environ.get(CHAIN_ENV)
    return parse_chain(spec) if spec else default_chain()


FAST_FAIL_S = 20.0      # an attempt that dies this fast with rc != 0 gets one retry on a fresh port


def free_port(host='127.0.0.1'):
    s = socket.socket()
    s.bind((host, 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _exited(pid):
    """True once process ``pid`` (our child) has terminated; it stays a zombie -- its pid and pgid reserved."""
    try:
        return os.waitid(os.P_PID, pid, os.WEXITED | os.WNOHANG | os.WNOWAIT) is not None
    except ChildProcessError:
        return True


def _kill_group(proc):
    """SIGKILL the process group this supervisor created for ``proc`` (start_new_session): the child and whatever it
    spawned, nothing else."""
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=10)
    except subprocess.TimeoutExpired:
        pass


def last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith('{'):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def run_attempt(cmd, env, budget_s, poll_s=0.2, peer_failed=None):
    """Run one child under a wall-clock budget.  Returns (status, stdout): status 'ok', 'rc=<n>', 'timeout' or
    'peer-failed' (``peer_failed()`` said another rank's child is already gone: this one can only wait for it)."""
    import tempfile
    with tempfile.TemporaryFile(mode='w+') as out:
        proc = subprocess.Popen(cmd, env=env, stdout=out, stderr=None, start_new_session=True)
        deadline = time.monotonic() + budget_s
        status, polls = None, 0
        while status is None:
            # look at the child WITHOUT reaping it (WNOWAIT): as long as the group leader is an un-reaped zombie its
            # pid -- and with it the process-group id killpg() is about to be given -- cannot be recycled
            polls += 1
            if _exited(proc.pid):
                _kill_group(proc)      # stragglers of a finished child (none expected); reaps the leader afterwards
                rc = proc.returncode
                if rc is None:             # _kill_group's bounded wait ran out before the leader was reaped
                    rc = proc.poll()
                status = 'ok' if rc == 0 else ('rc=unknown' if rc is None else 'rc=%d' % rc)
            elif time.monotonic() >= deadline:
                _kill_group(proc)
                status = 'timeout'
            elif peer_failed is not None and polls % 5 == 0 and peer_failed():
                _kill_group(proc)
                status = 'peer-failed'
            else:
                time.sleep(poll_s)
        out.seek(0)
        return status, out.read()


def supervise(script, argv, attempts=None, log=None):
    """Every launched rank calls this instead of running the benchmark itself.  Returns (line, tried) on rank 0 --
    ``line`` the winning attempt's parsed JSON (None if every transport failed) -- and (None, tried) elsewhere."""
    import torch.distributed as dist
    attempts = attempts or chain_from_env()
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    host = os.environ.get('MASTER_ADDR', '127.0.0.1')
    log = log or (lambda msg: sys.stderr.write('[bench supervisor rank %d] %s\n' % (rank, msg)))
    # the launcher's own rendezvous: under torch.distributed.run the agent already HOSTS a store on MASTER_PORT
    # (TORCHELASTIC_USE_AGENT_STORE) and every rank is a client; without an agent rank 0 hosts it -- env:// knows which
    base, _, _ = next(dist.rendezvous('env://', rank=rank, world_size=world,
                                      timeout=datetime.timedelta(seconds=120)))
    store = dist.PrefixStore('mvae_bench_supervisor', base)
    tried, winner = [], None
    for a0, att in enumerate(attempts):
      for retry in range(2):
        # free_port() closes its socket before the child's store binds the port, so another process can take it in
        # between: the attempt then dies within seconds as rc != 0 and the chain would fall back to a slower transport
        # for no reason.  Rank 0 grants ONE retry (fresh port) to an attempt that failed that fast.
        a = '%d.%d' % (a0, retry)
        if rank == 0:
            store.set('port/%s' % a, str(free_port(host)))
        child_port = int(store.get('port/%s' % a).decode())
        env = dict(os.environ)
        env.update(att.env)
        env[WORKER_ENV] = '1'
        env[TRANSPORT_ENV] = att.name
        env['MASTER_ADDR'] = host
        env['MASTER_PORT'] = str(child_port)
        env.pop('TORCHELASTIC_USE_AGENT_STORE', None)          # the child's rank 0 hosts its own store on ITS port
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL across processes needs on this driver
        t0 = time.monotonic()

        def peer_failed(a=a):
            try:
                for r in range(world):
                    key = 'out/%s/%d' % (a, r)
                    if r != rank and store.check([key]) and store.get(key).decode() != 'ok':
                        return True
            except Exception:
                return False
            return False
        status, stdout = run_attempt([sys.executable, script] + list(argv) + att.argv, env, att.budget_s,
                                     peer_failed=peer_failed)
        line = last_json_line(stdout) if rank == 0 else None
        if rank == 0 and status == 'ok' and line is None:
            status = 'no-line'
        store.set('out/%s/%d' % (a, rank), status)
        # the verdict is collective: every rank must have finished this attempt
        verdicts = []
        deadline = time.monotonic() + att.budget_s + 60.0
        for r in range(world):
            key = 'out/%s/%d' % (a, r)
            v = None
            while v is None and time.monotonic() < deadline:
                try:
                    if store.check([key]):
                        v = store.get(key).decode()
                    else:
                        time.sleep(0.2)
                except Exception:          # the store's host went away: nothing to agree with any more
                    v = 'store-lost'
            verdicts.append(v or 'silent')
        ok = all(v == 'ok' for v in verdicts)
        wall = time.monotonic() - t0
        tried.append({'transport': att.name, 'budget_s': att.budget_s, 'wall_s': round(wall, 1),
                      'ranks': verdicts if not ok else 'ok'})
        if retry:
            tried[-1]['retry'] = retry
        log('attempt %s (%s): %s in %.1f s' % (a, att.name, 'ok' if ok else verdicts, wall))
        if ok:
            winner = (att, line)
            break
        again = '0'
        try:
            if rank == 0:
                fast = retry == 0 and wall < FAST_FAIL_S and any(v.startswith('rc=') for v in verdicts)
                store.set('again/%s' % a, '1' if fast else '0')
            again = store.get('again/%s' % a).decode()
        except Exception:
            pass
        if again != '1':
            break
      if winner is not None:
        break
    # leave together: the store lives in rank 0's process
    try:
        store.set('bye/%d' % rank, '1')
        if rank == 0:
            deadline = time.monotonic() + 30.0
            while time.monotonic() < deadline and not store.check(['bye/%d' % r for r in range(world)]):
                time.sleep(0.1)
    except Exception:
        pass
    if rank != 0:
        return None, tried
    if winner is None:
        return None, tried
    att, line = winner
    dist = line.setdefault('dist', {}) if isinstance(line.get('dist', {}), dict) else {}
    dist.setdefault('transport', att.name)
    dist['transport_attempt'] = att.name
    dist['fallbacks_tried'] = tried[:-1]
    line['dist'] = dist
    return line, tried

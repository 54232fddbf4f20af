#!/usr/bin/env python
"""How long the HOST needs to enqueue one replayed step vs how long the GPU needs to run it (single
graph and the 3-graph data-parallel launch path at world size 1).  Tuning aid."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def run(kind, dp_mode):
    dev = torch.device('cuda', 0)
    model, eng, opt = bench.build(kind, bench.DEFAULT_BATCH[kind], dev, 1)
    dp = None
    if dp_mode:
        from mvae_amd.parallel import DataParallel
        dp = DataParallel(model, eng)
    batches = [bench.synthetic(kind, bench.DEFAULT_BATCH[kind], 1234 + i, dev) for i in range(4)]
    eng.capture(opt, batches[0][0].shape[1:], batches[0][1], comm=dp)
    for i in range(20):
        eng.replay(batches[i % 4][0], batches[i % 4][1], 0.5)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        eng.replay(batches[i % 4][0], batches[i % 4][1], 0.5)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('%s %s: host enqueue %.1f us/step, wall %.1f us/step' % (kind, '3-graph dp' if dp_mode else 'single graph',
                                                                    t_host / n * 1e6, t_all / n * 1e6))


if __name__ == '__main__' and len(sys.argv) == 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    for kind in ('mnist', 'celeba'):
        run(kind, False)
        run(kind, True)
    dist.destroy_process_group()


def breakdown(kind='mnist'):
    """Host time of each call of one data-parallel replay (world size 1)."""
    from mvae_amd.parallel import DataParallel
    dev = torch.device('cuda', 0)
    model, eng, opt = bench.build(kind, bench.DEFAULT_BATCH[kind], dev, 1)
    dp = DataParallel(model, eng)
    batches = [bench.synthetic(kind, bench.DEFAULT_BATCH[kind], 1234 + i, dev) for i in range(4)]
    eng.capture(opt, batches[0][0].shape[1:], batches[0][1], comm=dp)
    ga, gb, gc = eng._graphs
    acc = {}

    def timed(name, fn):
        t0 = time.perf_counter(); fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    n = 200
    for i in range(n + 20):
        if i == 20:
            torch.cuda.synchronize(); acc.clear()
        img, lbl = batches[i % 4]
        timed('bump+copies+coef', lambda: (eng._bump_bn_counters(), eng.static_image.copy_(img, non_blocking=True),
                                           eng.static_label.copy_(lbl, non_blocking=True), eng.set_coefficients(0.5)))
        timed('graph A', ga.replay)
        timed('allreduce 0', lambda: dp.launch(0))
        timed('graph B', gb.replay)
        timed('allreduce 1', lambda: dp.launch(1))
        timed('wait', dp.wait)
        timed('graph C', gc.replay)
    torch.cuda.synchronize()
    print('  '.join('%s %.1f us' % (k, v / n * 1e6) for k, v in acc.items()))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'breakdown':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    breakdown()
    dist.destroy_process_group()

#!/usr/bin/env python
"""Dump the per-call in-situ intervals bench.py's roofline is computed from (debug aid):
    python tools/insitu_dump.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mvae_amd.profiler import KernelProfile  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'mnist'
dev = torch.device('cuda:0')
batch = bench.DEFAULT_BATCH[kind]
model, eng, opt = bench.build(kind, batch, dev, 1)
batches = [bench.synthetic(kind, batch, 1234 + i, dev) for i in range(4)]
eng.side = eng.wg_main = eng.wg_side = None
for _ in range(2):
    eng.step(batches[0][0], batches[0][1], 0.5); opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.step(batches[0][0], batches[0][1], 0.5); opt.step()
host_s = time.perf_counter() - t0
torch.cuda.synchronize()
spin = int(bench._spin_cycles_per_second() * (2.5 * host_s * 3 + 0.005))
print('host enqueue per step %.3f ms, spin cycles %d' % (host_s * 1e3, spin))
with KernelProfile() as prof:
    t1 = time.perf_counter()
    torch.cuda._sleep(spin)
    for i in range(3):
        ta = time.perf_counter()
        eng.step(batches[i % 4][0], batches[i % 4][1], 0.5)
        opt.step()
        print('  step %d enqueued in %.3f ms' % (i, (time.perf_counter() - ta) * 1e3))
    print('enqueue total %.3f ms' % ((time.perf_counter() - t1) * 1e3))
    torch.cuda.synchronize()
    print('all done after %.3f ms' % ((time.perf_counter() - t1) * 1e3))
for name, key, flops, nbytes, e0, e1 in prof.records:
    print('%-28s %-22s %9.1f us' % (name, key, e0.elapsed_time(e1) * 1e3))

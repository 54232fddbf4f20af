#!/usr/bin/env python
"""Per-(launcher call, shape) kernel table of one train step from a rocprofv3 kernel trace.

rocprofv3 --stats aggregates by kernel TEMPLATE name, which mixes shapes (the MNIST step's dominant template has five
calls per step, four of one shape) -- so ``roofline.frac`` could not be recomputed from the committed summaries alone
(VERDICT r3).  Here every call of ``kernels.py`` in a single-stream eager step is preceded by an empty
``trace_marker_kernel`` (mvae_trace_marker): in the trace, ordered by start time, the k-th marker separates the kernels of
call k - 1 from those of call k, and the host side knows which (name, shape) call k was.

    rocprofv3 --kernel-trace -d gpurun_out/bs_mnist -o t -- python tools/step_by_shape.py run mnist gpurun_out/bs_mnist/calls.json
    python tools/step_by_shape.py collect gpurun_out/bs_mnist/calls.json <t_results.db> profiles/r04_by_shape.json \
        > profiles/r04_mnist_by_shape.txt

``collect`` prints the table (calls per step, average microseconds of the call's kernels = rocprof durations, the
kernels it launched, TFLOP/s or GB/s of the algorithmic work) and merges {"<workload>": {"<name> <key>": {...}}} into the
JSON table ``bench.py`` reads ``roofline.rocprof_avg_us`` from.
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_STEPS = 3


def run(kind, out_path):
    import torch
    import bench
    from mvae_amd.profiler import KernelProfile
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    batch = bench.DEFAULT_BATCH[kind]
    model, eng, opt = bench.build(kind, batch, dev, 1)
    batches = [bench.synthetic(kind, batch, 1234 + i, dev) for i in range(4)]
    eng.side = eng.wg_main = eng.wg_side = None          # one ordered queue: trace order = launch order
    for i in range(2):
        eng.step(batches[i][0], batches[i][1], 0.5); opt.step()
    torch.cuda.synchronize()
    with KernelProfile(marker=True, timed=False) as prof:
        for i in range(N_STEPS):
            eng.step(batches[i % 4][0], batches[i % 4][1], 0.5)
            opt.step()
        from mvae_amd import kernels as K
        K.trace_marker(-1)                               # closes the last call's segment
    torch.cuda.synchronize()
    with open(out_path, 'w') as f:
        json.dump({'workload': kind, 'batch': batch, 'steps': N_STEPS,
                   'calls': [list(c) for c in prof.sequence]}, f)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name.split('(')[0][:90]


def collect(calls_path, db_path, table_path):
    meta = json.load(open(calls_path))
    calls, steps = meta['calls'], meta['steps']
    c = sqlite3.connect(db_path)
    rows = c.execute('select start, end, name from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if 'trace_marker_kernel' in r[2]]
    if len(marks) != len(calls) + 1:
        raise SystemExit('%d markers in the trace, %d calls recorded (+1 closing marker expected)' % (len(marks), len(calls)))
    agg = {}
    for k, (name, key, flops, nbytes) in enumerate(calls):
        seg = rows[marks[k] + 1:marks[k + 1]]
        a = agg.setdefault((name, key), {'calls': 0, 'ns': 0, 'flops': flops, 'bytes': nbytes, 'kernels': {}})
        a['calls'] += 1
        a['ns'] += sum(e - s for s, e, _ in seg)
        for s, e, n in seg:
            a['kernels'][short(n)] = a['kernels'].get(short(n), 0) + 1
    print('# %s B=%d: one eager single-stream step, %d steps traced; rocprofv3 --kernel-trace durations per launcher call'
          % (meta['workload'], meta['batch'], steps))
    print('# columns: calls/step  avg_us (sum of the call\'s kernel durations)  us/step  TFLOP/s | GB/s (algorithmic, section 8d)  call  <- kernels')
    out = {}
    total = sum(a['ns'] for a in agg.values())
    for (name, key), a in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
        avg_us = a['ns'] / a['calls'] / 1e3
        rate = ''
        if a['flops']:
            rate = '%7.1f TF' % (a['flops'] / (avg_us * 1e-6) / 1e12)
        elif a['bytes']:
            rate = '%7.0f GB' % (a['bytes'] / (avg_us * 1e-6) / 1e9)
        kern = ', '.join('%s x%g' % (n, cnt / a['calls']) for n, cnt in a['kernels'].items())
        print('%6.2f %9.2f %9.1f  %10s  %-44s <- %s' % (a['calls'] / steps, avg_us, a['ns'] / steps / 1e3, rate,
                                                         ('%s %s' % (name, key))[:44], kern[:160]))
        out['%s %s' % (name, key)] = {'calls_per_step': a['calls'] / steps, 'rocprof_avg_us': round(avg_us, 3),
                                      'algorithmic_flops': a['flops'], 'algorithmic_bytes': a['bytes']}
    print('# sum of kernel durations per step: %.1f us in %d launcher calls' % (total / steps / 1e3, len(calls) // steps))
    table = {}
    if os.path.exists(table_path):
        table = json.load(open(table_path))
    table[meta['workload']] = out
    import mvae_amd  # noqa: F401
    from mvae_amd.profiler import code_stamp
    table.setdefault('_meta', {})[meta['workload']] = dict(code_stamp(), batch=meta.get('batch'))
    with open(table_path, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], sys.argv[3])
    else:
        collect(sys.argv[2], sys.argv[3], sys.argv[4])

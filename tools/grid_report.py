#!/usr/bin/env python
"""GPU box: block-count quantisation of every GEMM-shaped launch of one train step.

    MVAE_GRID_REPORT=1 MVAE_HIP_LIB=$PWD/multimodal-vae-public_amd/libmvae_hip_tuning.so \
        python tools/grid_report.py run celeba 2> grid_celeba.err
    python tools/grid_report.py table grid_celeba.err

`run` executes one eager step (the tuning build prints one `[grid]` line per launch: blocks, blocks per CU the kernel's
registers / LDS allow, rounds, the busiest CU's share); `table` prints the distinct launches sorted by what the quantisation
costs: work = blocks, capacity = 256 CUs x ceil(blocks / 256) block-times -- `balance` = work / capacity."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(kind):
    import torch
    import bench
    import mvae_amd
    from mvae_amd.engine import BimodalStep, Celeba19Step
    batch = bench.DEFAULT_BATCH[kind]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = getattr(mvae_amd, kind).model.MVAE(bench.N_LATENTS[kind]).to(dev).train()
    if kind == 'celeba19':
        eng = Celeba19Step(model, batch, 1.0, bench.LAMBDA_LABEL[kind], approx_m=1)
    else:
        eng = BimodalStep(model, batch, 1.0, bench.LAMBDA_LABEL[kind])
    image, label = bench.synthetic(kind, batch, 1, dev)
    sys.stderr.write('[grid] ---- step begins\n')
    eng.step(image, label, 0.5)
    torch.cuda.synchronize()


def table(path):
    rows = {}
    for line in open(path):
        if not line.startswith('[grid] tile'):
            continue
        m = re.match(r'\[grid\] tile (\d+)x(\d+) I (\d+) J (\d+) K (\d+)  blocks (\d+) \((\d+) x (\d+) x (\d+)\) threads (\d+) lds (\d+)  '
                     r'per_cu (\d+)  rounds (\d+)  busiest_cu (\d+)  balance ([0-9.]+)  items (\d+)  (.*)', line)
        if not m:
            continue
        g = m.groups()
        key = g[:12] + (g[15],)
        name = re.sub(r'.*\[with ', '', g[16])[:110]
        if key in rows:
            rows[key]['n'] += 1
        else:
            rows[key] = dict(n=1, tm=int(g[0]), tn=int(g[1]), I=int(g[2]), J=int(g[3]), K=int(g[4]), blocks=int(g[5]),
                             grid='%sx%sx%s' % g[6:9], threads=int(g[9]), per_cu=int(g[11]), rounds=int(g[12]),
                             busiest=int(g[13]), balance=float(g[14]), items=int(g[15]), name=name)
    out = sorted(rows.values(), key=lambda r: (1 - r['balance']) * 2.0 * r['I'] * r['J'] * r['K'] * r['n'], reverse=True)
    print('# n  GFLOP   tile    blocks (grid)        per_cu rounds busiest balance items  what')
    for r in out:
        gf = 2.0 * r['I'] * r['J'] * r['K'] / 1e9
        print('%2d %6.2f %4dx%-4d %6d %-14s %4d %5d %6d   %.3f %4d   I %d J %d K %d  %s'
              % (r['n'], gf, r['tm'], r['tn'], r['blocks'], r['grid'], r['per_cu'], r['rounds'], r['busiest'], r['balance'],
                 r['items'], r['I'], r['J'], r['K'], r['name']))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2])
    else:
        table(sys.argv[2])

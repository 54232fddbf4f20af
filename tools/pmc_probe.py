#!/usr/bin/env python
"""Run a few conv launches of the CelebA B=256 step for a rocprofv3 --pmc pass, or print the
per-dispatch counters of such a pass:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... -d out -o p -- python tools/pmc_probe.py run
    python tools/pmc_probe.py show out/.../p_results.db
"""
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import mvae_amd  # noqa: F401
    from gemm_bench import conv_cases, convT_cases
    B = 256
    cases = (conv_cases(B, 32, 32, 64, 2, 1, 'enc2') + conv_cases(B, 64, 16, 128, 2, 1, 'enc3')
             + convT_cases(2 * B, 128, 8, 64, 2, 1, 'dec2') + convT_cases(2 * B, 256, 5, 128, 1, 0, 'dec1'))
    for name, fl, fn in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print(name, '%.2f GFLOP' % (fl / 1e9))


def run_small():
    """The <= 4-channel conv launches (tools/small_conv_probe.py) once each, three repetitions."""
    import torch
    import mvae_amd  # noqa: F401
    from small_conv_probe import CASES
    for name, fl, fn in CASES:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print(name, '%.2f GFLOP' % (fl / 1e9))


def show(db):
    c = sqlite3.connect(db)
    rows = c.execute('select dispatch_id, kernel_name, counter_name, value from counters_collection '
                     'order by dispatch_id').fetchall()
    disp = {}
    for d, n, cn, v in rows:
        e = disp.setdefault(d, [n, {}])
        e[1][cn] = e[1].get(cn, 0.0) + v
    names = sorted({cn for _, n, cn, _ in rows})
    print('%-6s %-70s ' % ('id', 'kernel') + ' '.join('%16s' % n[-16:] for n in names))
    for d in sorted(disp):
        n, vals = disp[d]
        if not re.search(r'igemm|convT|conv_small|smallcin|conv_patch|wgrad_patch|gemm2', n):
            continue
        short = re.sub(r'\(anonymous namespace\)::|void ', '', n).split('(')[0][:70]
        print('%-6d %-70s ' % (d, short) + ' '.join('%16.0f' % vals.get(k, 0) for k in names))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        run()
    elif sys.argv[1] == 'run-small':
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        run_small()
    else:
        show(sys.argv[2])

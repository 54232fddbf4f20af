#!/usr/bin/env python
"""HBM traffic of one GEMM-shaped call, for bench.py's ``roofline.traffic``.

Two steps, both on the GPU box (MI355X_MICROARCH.md, HBM / rocprofv3 PMC sections: FETCH_SIZE and
WRITE_SIZE do not fit one pass, FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950):

    rocprofv3 --pmc FETCH_SIZE -d out/f -o p -- python tools/traffic_probe.py run linear_wgrad "M1024 N512 K512"
    rocprofv3 --pmc WRITE_SIZE -d out/w -o p -- python tools/traffic_probe.py run linear_wgrad "M1024 N512 K512"
    python tools/traffic_probe.py collect out/f/.../p_results.db out/w/.../p_results.db \
           profiles/r01_traffic.json linear_wgrad "M1024 N512 K512"

``run`` issues the call 10 times;
``collect`` sums FETCH_SIZE / WRITE_SIZE (KiB per dispatch) over the kernels ONE call launches
(GEMM + split finish + bias-gradient reduce), doubles the fetch figure, and merges
{"<name> <key>": {...}} into the JSON table bench.py reads.
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_CALLS = 10


def parse_key(key):
    return {m.group(1): int(m.group(2)) for m in re.finditer(r'([A-Z])(\d+)', key)}


def run(name, key):
    import torch
    import mvae_amd  # noqa: F401
    from mvae_amd import kernels as K
    dev = 'cuda'
    d = parse_key(key)
    M, N, Kd = d['M'], d['N'], d['K']
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    if name == 'linear_fwd':
        x, w, b, pre, act = r(M, Kd), r(N, Kd), r(N), torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: K.linear_fwd(x, w, b, pre, act)  # noqa: E731
    elif name == 'linear_dgrad':
        dy, w, dx, pre = r(M, N), r(N, Kd), torch.empty(M, Kd, device=dev), r(M, Kd)
        fn = lambda: K.linear_dgrad(dy, w, dx, pre)  # noqa: E731
    elif name == 'linear_wgrad':
        dy, x, dw, db = r(M, N), r(M, Kd), torch.empty(N, Kd, device=dev), torch.empty(N, device=dev)
        fn = lambda: K.linear_wgrad(dy, x, dw, db)  # noqa: E731
    else:
        raise SystemExit('unknown op %s' % name)
    for _ in range(N_CALLS):
        fn()
    torch.cuda.synchronize()


def _per_call_kib(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, value from counters_collection where counter_name = ?', (counter,)).fetchall()
    ours = [(n, v) for n, v in rows if re.search(r'igemm_kernel|finish|splitk_reduce', n)]
    per_kernel = {}
    for n, v in ours:
        short = re.sub(r'\(anonymous namespace\)::|void ', '', n).split('(')[0][:90]
        per_kernel[short] = per_kernel.get(short, 0.0) + v
    total = sum(v for _, v in ours)
    return total / N_CALLS, {k: v / N_CALLS for k, v in per_kernel.items()}


def collect(fetch_db, write_db, out_json, name, key):
    f_kib, f_detail = _per_call_kib(fetch_db, 'FETCH_SIZE')
    w_kib, w_detail = _per_call_kib(write_db, 'WRITE_SIZE')
    d = parse_key(key)
    M, N, Kd = d['M'], d['N'], d['K']
    algorithmic = 4 * (M * N + M * Kd + N * Kd + (N if name == 'linear_wgrad' else 0))
    ent = {
        'hbm_bytes_per_launch': int(round((2.0 * f_kib + w_kib) * 1024)),
        'fetch_size_kib_reported': round(f_kib, 1), 'fetch_correction': 2.0,
        'write_size_kib_reported': round(w_kib, 1),
        'algorithmic_bytes_per_launch': algorithmic,
        'kernels_fetch_kib': {k: round(v, 1) for k, v in f_detail.items()},
        'kernels_write_kib': {k: round(v, 1) for k, v in w_detail.items()},
        'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over %d launches; '
                  'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md)' % N_CALLS,
    }
    table = {}
    if os.path.exists(out_json):
        with open(out_json) as f:
            table = json.load(f)
    table['%s %s' % (name, key)] = ent
    with open(out_json, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(json.dumps(ent, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], sys.argv[3])
    else:
        collect(*sys.argv[2:7])

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
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    if name == 'linear_wgrad_batched':
        items = []
        for (M, N, Kd) in WGRAD_BATCHES[key]:
            items.append((r(M, N), r(M, Kd), torch.empty(N, Kd, device=dev), torch.empty(N, device=dev), False))
        for _ in range(N_CALLS):
            K.linear_wgrad_batched(items)
        torch.cuda.synchronize()
        return
    if (name, key) not in CONV_CASES:
        d = parse_key(key)
        M, N, Kd = d['M'], d['N'], d['K']
    if name == 'linear_fwd':
        x, w, b, pre, act = r(M, Kd), r(N, Kd), r(N), torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: K.linear_fwd(x, w, b, pre, act)  # noqa: E731
    elif name == 'linear_dgrad':
        dy, w, dx, pre = r(M, N), r(N, Kd), torch.empty(M, Kd, device=dev), r(M, Kd)
        fn = lambda: K.linear_dgrad(dy, w, dx, pre)  # noqa: E731
    elif name == 'linear_wgrad':
        dy, x, dw, db = r(M, N), r(M, Kd), torch.empty(N, Kd, device=dev), torch.empty(N, device=dev)
        fn = lambda: K.linear_wgrad(dy, x, dw, db)  # noqa: E731
    elif (name, key) in CONV_CASES:
        fn = conv_case(K, name, key)
    else:
        raise SystemExit('unknown op %s %s' % (name, key))
    for _ in range(N_CALLS):
        fn()
    torch.cuda.synchronize()


# the batched Linear weight-gradient launches of the MNIST step, by the profiler's key: (rows, out, in) per layer --
# the image decoder's four layers over its two terms (mnist/model.py:95-104 backward, batch 512)
WGRAD_BATCHES = {'4 layers': [(1024, 784, 512), (1024, 512, 512), (1024, 512, 512), (1024, 512, 64)]}

# conv launches bench.py may name as dominant on the CelebA step: (profiler name, key) -> ConvTranspose2d / Conv2d
# layer (B, Cin, H, Cout, stride, pad) of celeba/model.py:77-86,117-126 at 2 x 256 decoder rows / 256 encoder rows
CONV_CASES = {
    ('convT2d_dgrad', '512x256x5x5'): ('convT', 512, 256, 5, 128, 1, 0),
    ('convT2d_fwd', '512x128x8x8'): ('convT', 512, 256, 5, 128, 1, 0),
    ('convT2d_wgrad', '256x128x4x4'): ('convT', 512, 256, 5, 128, 1, 0),
    ('convT2d_fwd', '512x32x32x32'): ('convT', 512, 64, 16, 32, 2, 1),
    ('convT2d_wgrad', '128x64x4x4'): ('convT', 512, 128, 8, 64, 2, 1),      # the patch weight gradient (wgrad_patch.h)
    ('conv2d_fwd', '256x64x16x16'): ('conv', 256, 32, 32, 64, 2, 1),
    # FashionMNIST B = 1024 (two image-decoder terms): ConvTranspose2d(128, 64) on 7x7 maps, fashionmnist/model.py:112
    ('convT2d_dgrad', '2048x128x7x7'): ('convT', 2048, 128, 7, 64, 2, 1),
    ('convT2d_fwd', '2048x64x14x14'): ('convT', 2048, 128, 7, 64, 2, 1),
    # CelebA-19 B = 256: the 18 statistics-only decodes of ConvTranspose2d(256, 128) 5x5 -> 8x8, celeba19/model.py:143
    ('convT2d_fwd', '4608x128x8x8'): ('convT', 4608, 256, 5, 128, 1, 0),
    # ... and its ConvTranspose2d(128, 64) 8x8 -> 16x16: the 16x16 sibling of FashionMNIST's 7x7 -> 14x14 launch (L2 comparison)
    ('convT2d_fwd', '4608x64x16x16'): ('convT', 4608, 128, 8, 64, 2, 1),
}


def conv_case(K, name, key):
    import torch
    kind, B, Cin, H, Cout, s, p = CONV_CASES[(name, key)]
    r = lambda *sh: torch.randn(*sh, device='cuda')  # noqa: E731
    if kind == 'convT':
        OH = (H - 1) * s - 2 * p + 4
        x, w = r(B, Cin, H, H), r(Cin, Cout, 4, 4)
        y, dy, dx, dw = torch.empty(B, Cout, OH, OH, device='cuda'), r(B, Cout, OH, OH), torch.empty_like(x), torch.empty_like(w)
        return {'convT2d_fwd': lambda: K.convT2d_fwd(x, w, y, None, s, p),
                'convT2d_dgrad': lambda: K.convT2d_dgrad(dy, w, dx, None, s, p),
                'convT2d_wgrad': lambda: K.convT2d_wgrad(dy, x, dw, s, p)}[name]
    OH = (H + 2 * p - 4) // s + 1
    x, w = r(B, Cin, H, H), r(Cout, Cin, 4, 4)
    y, dy, dx, dw = torch.empty(B, Cout, OH, OH, device='cuda'), r(B, Cout, OH, OH), torch.empty_like(x), torch.empty_like(w)
    return {'conv2d_fwd': lambda: K.conv2d_fwd(x, w, y, None, s, p),
            'conv2d_dgrad': lambda: K.conv2d_dgrad(dy, w, dx, None, s, p),
            'conv2d_wgrad': lambda: K.conv2d_wgrad(dy, x, dw, s, p)}[name]


def algorithmic_bytes(name, key):
    """Bytes a launch must move at least: every operand read once, the result written once."""
    if (name, key) in CONV_CASES:
        kind, B, Cin, H, Cout, s, p = CONV_CASES[(name, key)]
        OH = (H - 1) * s - 2 * p + 4 if kind == 'convT' else (H + 2 * p - 4) // s + 1
        return 4 * (B * Cin * H * H + B * Cout * OH * OH + Cin * Cout * 16)
    if name == 'linear_wgrad_batched':
        # x of layer l + 1 is the activation behind dy of layer l's producer: distinct tensors, each read once
        return sum(4 * (M * N + M * Kd + N * Kd + N) for (M, N, Kd) in WGRAD_BATCHES[key])
    d = parse_key(key)
    M, N, Kd = d['M'], d['N'], d['K']
    return 4 * (M * N + M * Kd + N * Kd + (N if name == 'linear_wgrad' else 0))


def _per_call_kib(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, value from counters_collection where counter_name = ?', (counter,)).fetchall()
    ours = [(n, v) for n, v in rows if re.search(r'igemm_kernel|gemm2s?_kernel|g2_finish|dgrad_smalln|finish|convT_s1|convT_small|conv_small|convT_patch|conv_patch|wgrad_patch|wgrad_direct|wgrad_batched|wgrad_smallcin|repack_dgrad', n)]
    per_kernel = {}
    for n, v in ours:
        short = re.sub(r'\(anonymous namespace\)::|void ', '', n).split('(')[0][:90]
        per_kernel[short] = per_kernel.get(short, 0.0) + v
    total = sum(v for _, v in ours)
    return total / N_CALLS, {k: v / N_CALLS for k, v in per_kernel.items()}


def collect(fetch_db, write_db, out_json, name, key):
    f_kib, f_detail = _per_call_kib(fetch_db, 'FETCH_SIZE')
    w_kib, w_detail = _per_call_kib(write_db, 'WRITE_SIZE')
    algorithmic = algorithmic_bytes(name, key)
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
    import mvae_amd  # noqa: F401
    from mvae_amd.profiler import code_stamp
    ent['collected_on'] = code_stamp()
    table['%s %s' % (name, key)] = ent
    with open(out_json, 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(json.dumps(ent, indent=1))


def counters(db):
    """Any --pmc pass over ``run``: every counter, per launch, of the kernels one call launches (L2 hit / miss, fabric requests)."""
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, counter_name, value from counters_collection').fetchall()
    acc = {}
    for n, cn, v in rows:
        if re.search(r'igemm_kernel|gemm2s?_kernel|g2_finish|dgrad_smalln|finish|convT_s1|convT_small|conv_small|convT_patch|conv_patch|wgrad_patch|wgrad_direct|wgrad_batched|wgrad_smallcin', n):
            short = re.sub(r'\(anonymous namespace\)::|void ', '', n).split('(')[0][:100]
            acc.setdefault(short, {}).setdefault(cn, 0.0)
            acc[short][cn] += v / N_CALLS
    for k, d in acc.items():
        print(k)
        for cn in sorted(d):
            print('    %-26s %14.0f' % (cn, d[cn]))
        if 'TCC_HIT_sum' in d and 'TCC_MISS_sum' in d:
            print('    %-26s %14.3f' % ('L2 hit rate', d['TCC_HIT_sum'] / max(1.0, d['TCC_HIT_sum'] + d['TCC_MISS_sum'])))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'counters':
        counters(sys.argv[2])
    else:
        collect(*sys.argv[2:7])

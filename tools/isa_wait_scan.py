#!/usr/bin/env python
"""Where does a kernel wait for a load it has only just issued?

hipcc (ROCm 7.2) drains the memory queue behind a load that sits under a branch -- `if (ptr) v += ptr[i]` in an
unrolled epilogue becomes branch, load, `s_waitcnt vmcnt(0)`, and on gfx9 that counter also covers earlier stores --
so a thread's outputs turn into a chain of dependent memory round trips (4-8 us of an 11-us GEMM
launch).  This scans the ISA of every kernel of a csrc/*.hip file for the signature: a vector memory load followed
within a few instructions by a wait that leaves (almost) nothing outstanding, and counts the occurrences, apart for
those inside a loop body (once per trip) and in straight-line code.

    python tools/isa_wait_scan.py linear.hip [filter-substring] [extra hipcc flags]
"""
import os
import re
import subprocess
import sys
from collections import Counter

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-vae-public_amd', 'csrc')
NEAR = 3          # instructions between the load and the wait
LEFT = 1          # the wait leaves at most this many operations outstanding


def scan(fn):
    lines = [l.strip() for l in fn.split('\n')]
    ops = [(i, l) for i, l in enumerate(lines) if l and not l.startswith(';') and not l.startswith('.')
           and not l.endswith(':') or re.match(r'^\.LBB\d+_\d+:', l)]
    labels = {m.group(1): k for k, (_, l) in enumerate(ops) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    in_loop = [False] * len(ops)
    for k, (_, l) in enumerate(ops):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            for q in range(labels[m.group(1)], k + 1):
                in_loop[q] = True
    hits = Counter()
    n_loads = 0
    for k, (_, l) in enumerate(ops):
        if not re.match(r'(global_load|buffer_load|flat_load)', l):
            continue
        n_loads += 1
        for q in range(k + 1, min(k + 1 + NEAR, len(ops))):
            m = re.search(r'vmcnt\((\d+)\)', ops[q][1])
            if ops[q][1].startswith('s_waitcnt') and m and int(m.group(1)) <= LEFT:
                hits['loop' if in_loop[k] else 'straight'] += 1
                break
            if re.match(r'(global_load|buffer_load|flat_load)', ops[q][1]):
                break
    return n_loads, hits


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
    extra = [a for a in sys.argv[2:] if a.startswith('-')]
    asm = '/tmp/isa_wait_%d.s' % os.getpid()
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S',
                    '--cuda-device-only'] + extra + [src, '-o', asm], cwd=CSRC, stderr=subprocess.DEVNULL, check=True)
    text = open(asm).read()
    os.remove(asm)
    rows = []
    for fn in re.split(r'\n(?=_Z\w+:)', text):
        m = re.match(r'(_Z\w+):', fn)
        if not m:
            continue
        name = subprocess.run(['c++filt', m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = name.replace('(anonymous namespace)::', '').replace('void ', '', 1)
        name = name[:name.index('>(') + 1] if '>(' in name else name.split('(')[0]
        if flt not in name:
            continue
        n_loads, hits = scan(fn)
        if hits:
            rows.append((hits['straight'] + hits['loop'], name, n_loads, hits['straight'], hits['loop']))
    print('%-120s %6s %9s %6s   (load followed within %d instructions by s_waitcnt vmcnt(<= %d))' % (
        'kernel', 'loads', 'straight', 'loop', NEAR, LEFT))
    for _, name, n_loads, a, b in sorted(rows, reverse=True):
        print('%-120s %6d %9d %6d' % (name[:120], n_loads, a, b))


if __name__ == '__main__':
    main()

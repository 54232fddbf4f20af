#!/usr/bin/env python
"""Instruction mix of the MFMA loops of every igemm kernel of a csrc/*.hip file: VALU instructions per MFMA is
the number to watch on fp32 MFMA (each VALU instruction costs matrix throughput, tools/mfma_peak.hip).
    python tools/loop_mix.py conv.hip [filter-substring] [extra hipcc flags]"""
import os
import re
import subprocess
import sys
from collections import Counter

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-vae-public_amd', 'csrc')


def category(op):
    if 'mfma' in op: return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if 'load' in op or 'store' in op: return 'vmem'
    return 'other'


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
    extra = [a for a in sys.argv[2:] if a.startswith('-')]
    asm = '/tmp/loop_mix_%d.s' % os.getpid()
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S',
                    '--cuda-device-only'] + extra + [src, '-o', asm], cwd=CSRC, stderr=subprocess.DEVNULL, check=True)
    text = open(asm).read()
    os.remove(asm)
    funcs = re.split(r'\n(?=_Z\w+:)', text)
    for fn in funcs:
        m = re.match(r'(_Z\w+):', fn)
        if not m:
            continue
        name = subprocess.run(['c++filt', m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = name.replace('(anonymous namespace)::', '').replace('void ', '', 1)
        name = name[:name.index('>(') + 1] if '>(' in name else name.split('(')[0]
        if flt not in name:
            continue
        lines = fn.split('\n')
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r'^(\.LBB\d+_\d+):', l)
            if mm:
                labels[mm.group(1)] = i
        for i, l in enumerate(lines):
            mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
            if not (mm and mm.group(1) in labels and labels[mm.group(1)] < i):
                continue
            body = lines[labels[mm.group(1)]:i + 1]
            c = Counter()
            for b in body:
                t = b.strip().split()
                if not t or t[0].startswith(';') or t[0].endswith(':') or t[0].startswith('.'):
                    continue
                c[category(t[0])] += 1
            if c['mfma'] < 8:
                continue
            print('%-92s mfma %3d valu %3d (%.2f/mfma) lds %3d vmem %3d salu %3d wait %2d' % (
                name[:92], c['mfma'], c['valu'], c['valu'] / c['mfma'], c['lds'], c['vmem'], c['salu'], c['wait']))


if __name__ == '__main__':
    main()

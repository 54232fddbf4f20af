#!/usr/bin/env python
"""VGPRs / spills / occupancy of every kernel of a csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py conv.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-vae-public_amd', 'csrc')


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + extra + \
          ['-c', src, '-o', '/tmp/kr_%d.o' % os.getpid(), '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith('Function Name:'):
            cur = {'name': t.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.rsplit(':', 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows),
                           stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        n = n.replace('(anonymous namespace)::', '').replace('void ', '', 1)
        n = n.split('(')[0] if '>(' not in n else n[:n.index('>(') + 1]
        print('%-120s vgpr %3s spill %3s scratch %4s occ %s lds %s' % (
            n[:120], r.get('VGPRs'), r.get('VGPRs Spill'), r.get('ScratchSize [bytes/lane]'),
            r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
    os.remove('/tmp/kr_%d.o' % os.getpid())


if __name__ == '__main__':
    main()

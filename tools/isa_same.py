#!/usr/bin/env python
"""Did a source change touch the code of kernels it was not meant to touch?  Compile a csrc/*.hip file to gfx950
assembly at two git revisions (or the working tree) and compare every kernel instruction by instruction (labels
renumbered, comments dropped):
    python tools/isa_same.py conv.hip HEAD~1            # HEAD~1 vs working tree
    python tools/isa_same.py linear.hip v1 v2
Prints the kernels that differ, the ones only one side has, and the count of identical ones."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join('multimodal-vae-public_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-S', '--cuda-device-only']


def assembly(src, rev):
    if rev is None:
        tree = ROOT
    else:
        tree = tempfile.mkdtemp(prefix='isa_')
        subprocess.run('git archive %s %s include | tar -x -C %s' % (rev, CSRC, tree), shell=True, check=True, cwd=ROOT)
    out = os.path.join(tempfile.mkdtemp(prefix='isa_'), 'k.s')
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [src, '-o', out], cwd=os.path.join(tree, CSRC), check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text):
    out = {}
    for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:', text, flags=re.S | re.M):
        body = re.sub(r'\.LBB\d+_', '.LBB_', m.group(2))
        body = re.sub(r'\.Ltmp\d+', '.Ltmp', body)
        lines = [re.sub(r'\s*;.*$', '', l) for l in body.splitlines()
                 if not l.strip().startswith(('.loc', '.file', ';', '.cfi'))]
        out[m.group(1)] = hashlib.md5('\n'.join(lines).encode()).hexdigest()
    return out


def main():
    src = sys.argv[1]
    a = sys.argv[2]
    b = sys.argv[3] if len(sys.argv) > 3 else None
    ka, kb = kernels(assembly(src, a)), kernels(assembly(src, b))
    names = subprocess.run(['c++filt'], input='\n'.join(sorted(set(ka) | set(kb))), stdout=subprocess.PIPE,
                           text=True).stdout.splitlines()
    pretty = dict(zip(sorted(set(ka) | set(kb)), names))
    same = [k for k in ka if kb.get(k) == ka[k]]
    for k in ka:
        if k in kb and kb[k] != ka[k]:
            print('DIFFERENT  %s' % pretty[k][:200])
    for k in ka:
        if k not in kb:
            print('ONLY %s  %s' % (a, pretty[k][:200]))
    for k in kb:
        if k not in ka:
            print('ONLY %s  %s' % (b or 'worktree', pretty[k][:200]))
    print('%d kernels identical, %d in %s, %d in %s' % (len(same), len(ka), a, len(kb), b or 'worktree'))


if __name__ == '__main__':
    main()

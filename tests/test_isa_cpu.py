"""Properties of the compiled kernels that the step time depends on, checked on the ISA hipcc emits (no GPU needed).

DESIGN.md 5.7: an operand load under a block-uniform branch in an epilogue (`if (bias) v += bias[j]`) makes hipcc wait for
the whole memory queue behind it, and a thread's outputs become a chain of dependent memory round trips -- 4-8 us of an
11-us launch in the small (k-grouped) GEMM layouts until their operands were fetched ahead of the main loop.  This keeps
it that way: behind the last matrix instruction those kernels may load nothing but the optional accumulate operand.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'multimodal-vae-public_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which('c++filt') is None,
                                reason='needs hipcc and c++filt')


@pytest.fixture(scope='module')
def linear_kernels(tmp_path_factory):
    asm = str(tmp_path_factory.mktemp('isa') / 'linear.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only',
                    'linear.hip', '-o', asm], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read()
    out = {}
    chunks = re.split(r'\n(?=_Z\w+:)', text)
    names = subprocess.run(['c++filt'], input='\n'.join(re.match(r'(_Z\w+):', c).group(1) if re.match(r'(_Z\w+):', c) else '-'
                                                         for c in chunks), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for chunk, name in zip(chunks, names):
        if 'igemm_kernel<' not in name:
            continue
        name = name.replace('(anonymous namespace)::', '')
        name = name[name.index('igemm_kernel<'):name.index('>(') + 1]
        out[name] = [l.strip() for l in chunk.split('\n')]
    return out


def _loads_behind_last_mfma(lines):
    last = max(i for i, l in enumerate(lines) if 'v_mfma' in l)
    return sum(1 for l in lines[last:] if re.match(r'(global_load|buffer_load|flat_load)', l))


def _loads_ahead_of_first_mfma(lines):
    first = min(i for i, l in enumerate(lines) if 'v_mfma' in l)
    return sum(1 for l in lines[:first] if re.match(r'(global_load|buffer_load|flat_load)', l))


# the MNIST step's Linear launches (mnist/model.py:67-146 at batch 512: profiles/r04_mnist_by_shape.txt): per kernel the
# outputs a thread finishes in the cooperative epilogue and the (static) tile loads ahead of the first matrix instruction
SMALL_LAYOUTS = [
    ('igemm_kernel<LdRowsKT<32, true, 64>, LdRowsKT<64, true, 64>, EpRowMajor, 1, 1, false, 4, 1, 2>', 4, 18),     # forward 32 x 64
    ('igemm_kernel<LdRowsKT<32, true, 64>, LdRowsMNT<64, true, 64>, EpRowMajor, 1, 1, false, 4, 1, 2>', 4, 18),    # data gradient
    ('igemm_kernel<LdRowsKT<32, true, 64>, LdRowsKT<32, true, 64>, EpRowMajor, 1, 1, false, 8, 1, 1>', 2, 12),     # 32 x 32 tiles
    ('igemm_kernel<LdRowsKT<32, true, 64>, LdRowsMNT<32, true, 64>, EpRowMajor, 1, 1, false, 8, 1, 1>', 2, 12),
]


@pytest.mark.parametrize('kernel,outputs,tile_loads', SMALL_LAYOUTS)
def test_small_layout_epilogue_loads_only_the_accumulate_operand(linear_kernels, kernel, outputs, tile_loads):
    lines = linear_kernels[kernel]
    # one optional load per output is left behind the reduction: `if (accumulate) v += out[idx]`
    assert _loads_behind_last_mfma(lines) <= outputs, kernel
    # ... and the pre-activation / mask (and bias) of every output were requested before the first matrix instruction.
    # (Since MVAE_PHASED_PRELOAD=2 the MFMA-only waves' load-free copy of the loop comes first in the text, so the movers'
    # first tile loads are no longer textually ahead of the first v_mfma: what IS ahead of it is exactly the epilogue's
    # operand prefetch, issued before the waves part ways; the tile loads are counted over the whole kernel.)
    assert _loads_ahead_of_first_mfma(lines) >= 2 * outputs, kernel
    total = sum(1 for l in lines if re.match(r'(global_load|buffer_load|flat_load)', l))
    assert total >= tile_loads + 2 * outputs, kernel


@pytest.mark.parametrize('epilogue', ['EpRowBce', 'EpRowCe'])
def test_loss_folding_epilogues_load_nothing_behind_the_reduction(linear_kernels, epilogue):
    names = [n for n in linear_kernels if epilogue in n and re.search(r'false, [248], [12], [12]>$', n)
             and not re.search(r'false, [248], 2, 2>$', n)]
    assert names, 'no k-grouped %s kernel found' % epilogue
    for n in names:
        per_thread = {'4, 2, 1': 4, '4, 1, 2': 4, '2, 2, 1': 8, '2, 1, 2': 8, '8, 1, 1': 2, '4, 1, 1': 4}[n[-8:-1]]
        if per_thread <= 8:
            assert _loads_behind_last_mfma(linear_kernels[n]) == 0, n

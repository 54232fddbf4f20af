"""CPU: the C-ABI library loads and exports every symbol include/mvae_hip.h declares; host-side
plan compilation, arena layout and error behaviour (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

import mvae_amd
from mvae_amd import _lib, layers as L
from mvae_amd.arena import ParamArena

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(tuning=False):
    """Functions include/mvae_hip.h declares: the product ABI, or (tuning=True) the block under
    ``#ifdef MVAE_TUNING`` that only libmvae_hip_tuning.so exports."""
    text = open(os.path.join(ROOT, 'include', 'mvae_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    m = re.search(r'#ifdef MVAE_TUNING(.*?)#endif', text, flags=re.S)
    assert m, 'the tuning block is missing from the header'
    text = m.group(1) if tuning else text.replace(m.group(0), '')
    return sorted(set(re.findall(r'\b(mvae_[a-zA-Z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'libmvae_hip.so is not built: run __graft_entry__.build()'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(handle, name), 'header declares %s but the library does not export it' % name
    assert handle.mvae_abi_version() == 2


def test_product_library_has_no_tuning_state():
    """include/mvae_hip.h: 're-entrant: no global mutable state' -- the mvae_debug_* overrides live only in
    the -DMVAE_TUNING build."""
    handle = ctypes.CDLL(_lib.LIB_PATH)
    tuning = _header_functions(tuning=True)
    assert tuning and all(n.startswith('mvae_debug_') for n in tuning)
    for name in tuning:
        assert not hasattr(handle, name), 'libmvae_hip.so exports the tuning hook %s' % name
    assert sorted(_lib._TUNING_SIGNATURES) == tuning
    if os.path.exists(_lib.TUNING_LIB_PATH):
        th = ctypes.CDLL(_lib.TUNING_LIB_PATH)
        for name in tuning + _header_functions():
            assert hasattr(th, name), 'tuning build lacks %s' % name


def test_binding_table_matches_header():
    assert _lib.exported_symbols() == _header_functions()
    _lib.lib()   # sets argtypes for every entry; AttributeError = mismatch


def test_workspace_queries_are_pure_host_calls():
    lib = _lib.lib()
    assert lib.mvae_gemm_ws_bytes(512, 784, 1024) > 0
    assert lib.mvae_gemm_ws_bytes(0, 784, 1024) == 0
    assert lib.mvae_bn_ws_bytes(3, 64, 256 * 256) == 3 * 64 * 10 * 3 * 4


def test_bad_arguments_return_error_codes_without_a_gpu():
    lib = _lib.lib()
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.mvae_linear_fwd(None, 4, None, None, None, None, 4, None, 1.0, 4, 4, 4, None, 0, None) == -1
    assert lib.mvae_conv2d_k4_fwd(None, None, None, None, 1, 1, 8, 8, 1, 3, 1, None) == -1
    assert lib.mvae_fill(None, 4, 0.0, None) == -1
    # the entry points added in round 2: null pointers / bad item lists are refused before anything is launched
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.mvae_adam_apply_at(None, p, p, p, 16, 1e-3, 0.9, 0.999, 1e-8, 1.0, p, 0, None) == -1
    assert lib.mvae_adam_apply_at(p, p, p, p, 16, 1e-3, 0.9, 0.999, 1e-8, 1.0, None, 0, None) == -1
    assert lib.mvae_adam_apply_at(p, p, p, p, 0, 1e-3, 0.9, 0.999, 1e-8, 1.0, p, 0, None) == 0        # nothing to do
    assert lib.mvae_linear_wgrad_batched(None, 1, None) == -1
    assert lib.mvae_conv_k4_repack_batched(None, 1, None) == -1
    assert lib.mvae_conv2d_k4_fwd_stats(p, p, p, 8, 32, 32, 32, 64, 2, 1, None, 0, None, None) == -1   # no record buffer
    lay = _lib.StatsLayout()
    assert lib.mvae_conv_k4_stats_layout(0, 8, 32, 32, 32, 64, 3, 1, ctypes.byref(lay)) == -1           # stride 3
    assert lib.mvae_conv_k4_stats_layout(0, 8, 32, 32, 32, 64, 2, 1, None) == -1


def test_no_cpu_fallback():
    m = mvae_amd.mnist.model.MVAE(8)
    with pytest.raises(RuntimeError, match='GPU'):
        m(torch.zeros(2, 1, 28, 28), torch.zeros(2, dtype=torch.long))
    with pytest.raises(RuntimeError, match='GPU'):
        m.finalize()
    with pytest.raises(RuntimeError, match='GPU'):
        mvae_amd.functional.elbo_loss_label(None, None, None, None, torch.zeros(2, 8), torch.zeros(2, 8))


@pytest.mark.parametrize('kind,n_latents,n_params', [
    ('mnist', 64, 2588186), ('fashionmnist', 64, 7689610), ('celeba', 100, 6373346),
    ('celeba19', 100, 22395690)])
def test_state_dict_keys_and_sizes_match_reference(kind, n_latents, n_params):
    """parameter counts dumped from the reference modules; key sets are
    compared with the oracle modules, which load into the real reference in make_golden.py."""
    from oracle import models as OM
    model = getattr(mvae_amd, kind).model.MVAE(n_latents)
    assert sum(p.numel() for p in model.parameters()) == n_params
    ref = OM.MODELS[kind][0](n_latents).state_dict()
    mine = model.state_dict()
    assert set(ref.keys()) == set(mine.keys())
    for k in ref:
        assert tuple(ref[k].shape) == tuple(mine[k].shape), k


def test_plan_compilation_fuses_the_reference_layer_patterns():
    c = mvae_amd.celeba.model.MVAE(100)
    enc = [(op.kind, op.act, op.drop) for op in c.image_encoder.plan()]
    assert enc == [('conv', True, 0.0), ('conv', False, 0.0), ('bn', True, 0.0), ('conv', False, 0.0),
                   ('bn', True, 0.0), ('conv', False, 0.0), ('bn', True, 0.0), ('view', False, 0.0),
                   ('lin', True, 0.1), ('lin', False, 0.0)]
    dec = [(op.kind, op.act) for op in c.image_decoder.plan()]
    assert dec == [('lin', True), ('view', False), ('convT', False), ('bn', True), ('convT', False),
                   ('bn', True), ('convT', False), ('bn', True), ('convT', False)]
    m = mvae_amd.mnist.model.MVAE(64)
    assert [op.kind for op in m.text_encoder.plan()] == ['emb', 'lin', 'lin2']
    with pytest.raises(RuntimeError, match='4x4'):
        L.compile_plan([L.Conv2d(3, 8, 3, 1, 1, bias=False)])


def test_arena_layout_on_cpu_tensors():
    """The arena itself is device-agnostic bookkeeping; check ordering/adjacency/alignment."""
    m = mvae_amd.mnist.model.MVAE(64)
    arena = ParamArena(m, order=m.arena_order(), adjacent=m.arena_adjacent())
    assert arena.numel >= 2588186
    enc = m.image_encoder
    assert enc.fc31.weight._arena_off + enc.fc31.weight.numel() == enc.fc32.weight._arena_off
    assert enc.fc31.bias._arena_off + enc.fc31.bias.numel() == enc.fc32.bias._arena_off
    w, _ = arena.joined(enc.fc31.weight, enc.fc32.weight)
    assert w.shape == (128, 512)
    assert torch.equal(w[:64], enc.fc31.weight) and torch.equal(w[64:], enc.fc32.weight)
    for p in arena.params:
        assert p.data_ptr() == arena.flat.data_ptr() + 4 * p._arena_off
    # decoders first (backward-completion order), every range inside the arena
    lo_d, hi_d = arena.module_ranges[m.image_decoder]
    lo_e, hi_e = arena.module_ranges[m.image_encoder]
    assert lo_d == 0 and hi_d <= lo_e
    from mvae_amd.parallel import bucket_ranges
    (a0, a1), (b0, b1) = bucket_ranges(m, arena)
    assert a0 == 0 and a1 == b0 and b1 == arena.numel
    # state_dict round trip keeps the views
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert enc.fc1.weight.data_ptr() == arena.flat.data_ptr() + 4 * enc.fc1.weight._arena_off


# ---- statistics launches (conv epilogue -> BatchNorm): the layout query and the argument checks are host logic ----
CELEBA_STATS_SHAPES = [
    # transposed, B, Cin, H, W, Cout, stride, pad, has_layout       (celeba/model.py:79-86,117-126 at B = 256, 3 decodes)
    (0, 256, 3, 64, 64, 32, 2, 1, False),        # first conv: direct small-channel kernel (and no BatchNorm behind it)
    (0, 256, 32, 32, 32, 64, 2, 1, True),
    (0, 256, 64, 16, 16, 128, 2, 1, True),
    (0, 256, 128, 8, 8, 256, 1, 0, True),
    (1, 768, 256, 1, 1, 128, 1, 0, False),       # 1x1 -> 4x4 transposed conv: its own stride-1 kernel
    (1, 768, 128, 4, 4, 64, 2, 1, True),
    (1, 768, 64, 8, 8, 32, 2, 1, True),
    (1, 768, 32, 16, 16, 3, 2, 1, False),        # last layer: direct kernel
    (1, 19 * 256, 64, 8, 8, 32, 2, 1, True),     # celeba19: 19 decodes in one launch
    (0, 7, 128, 8, 8, 256, 1, 0, False),         # 7 * 25 positions: a ragged last tile
]


@pytest.mark.parametrize('tr,B,Cin,H,W,Cout,s,p,has', CELEBA_STATS_SHAPES)
def test_stats_layout_covers_every_output_position_once(tr, B, Cin, H, W, Cout, s, p, has):
    from mvae_amd import kernels as K
    lay = K.conv_stats_layout(tr, B, Cin, H, W, Cout, s, p)
    assert (lay is not None) == has
    if lay is None:
        return
    OH = (H - 1) * s - 2 * p + 4 if tr else (H + 2 * p - 4) // s + 1
    OW = (W - 1) * s - 2 * p + 4 if tr else (W + 2 * p - 4) // s + 1
    assert lay.ncls == (s * s if tr else 1)
    assert lay.parts() * lay.cols == B * OH * OW
    assert lay.cols in (32, 64) and lay.ppt in (1, 2, 4)


def test_bn_from_parts_rejects_layouts_that_do_not_match_the_tensor():
    lib = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    G, B, C, HW = 3, 256, 64, 256

    def call(lay, x=p, y=p):
        return lib.mvae_bn_train_fwd_parts(x, p, p, y, p, p, None, None, G, B, C, HW, 1e-5, 0.1, 1, None, 0, p,
                                           ctypes.byref(lay), p, 1 << 30, None)
    ERR_ARG = -1
    assert _lib.StatsLayout(4, 768, 2, 32).parts() * 32 == G * B * HW       # the layout that WOULD be accepted
    assert call(_lib.StatsLayout(4, 768, 2, 32), x=None, y=p) == ERR_ARG    # an output needs x
    assert call(_lib.StatsLayout(4, 192, 2, 32)) == ERR_ARG                 # covers a quarter of the tensor
    assert call(_lib.StatsLayout(4, 767, 2, 32)) == ERR_ARG                 # wrong count
    assert call(_lib.StatsLayout(2, 1024, 3, 32)) == ERR_ARG                # right count, tiles straddle groups
    assert call(_lib.StatsLayout(0, 0, 0, 0)) == ERR_ARG

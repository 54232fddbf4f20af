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
    assert handle.mvae_abi_version() == 6


def test_shared_memory_nccl_stand_in_exports_what_the_communicator_binds():
    """tests/shm_nccl (test infrastructure for the world-2 GPU test): every symbol csrc/comm.hip resolves with dlsym."""
    path = os.path.join(ROOT, 'tests', 'shm_nccl', 'libshm_nccl.so')
    assert os.path.exists(path), 'libshm_nccl.so is not built: run __graft_entry__.build()'
    text = open(os.path.join(ROOT, 'multimodal-vae-public_amd', 'csrc', 'comm.hip')).read()
    wanted = sorted(set(re.findall(r'"(nccl[A-Za-z]+)"', text)))
    assert 'ncclAllReduce' in wanted and 'ncclCommGetAsyncError' in wanted and 'ncclReduceScatter' in wanted and len(wanted) == 10
    handle = ctypes.CDLL(path)
    for name in wanted:
        assert hasattr(handle, name), name
    v = ctypes.c_int(0)
    assert handle.ncclGetVersion(ctypes.byref(v)) == 0 and v.value == 10000


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


def test_reference_module_level_names_exist_on_every_drop_in():
    """``from model import MVAE, ProductOfExperts, Swish, prior_expert`` and ``model.experts`` -- the names the
    reference's model.py defines (mnist/model.py:14,149,166,172; celeba/model.py:13,193,210,216)."""
    import re
    header = open(os.path.join(ROOT, 'include', 'mvae_hip.h')).read()
    assert int(re.search(r'#define\s+MVAE_POE_NO_PRIOR\s+(\d+)', header).group(1)) == _lib.POE_NO_PRIOR
    for kind, variant in (('mnist', 'A'), ('fashionmnist', 'A'), ('celeba', 'B'), ('celeba19', 'B')):
        mod = getattr(mvae_amd, kind).model
        for name in ('MVAE', 'ProductOfExperts', 'Swish', 'prior_expert'):
            assert hasattr(mod, name), '%s/model.py lacks %s' % (kind, name)
        assert mod.ProductOfExperts.VARIANT == variant
        m = mod.MVAE(8)
        assert isinstance(m.experts, mod.ProductOfExperts)
        assert not [k for k in m.state_dict() if k.startswith('experts')]      # parameter-free: keys unchanged
        mu, lv = mod.prior_expert((1, 3, 8))
        assert mu.shape == (1, 3, 8) and not mu.is_cuda and float(mu.abs().sum() + lv.abs().sum()) == 0.0
        with pytest.raises(RuntimeError, match='GPU'):
            m.experts(mu, lv)
    # the C ABI refuses an empty product
    ex = _lib.Experts()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert _lib.lib().mvae_poe_fwd(ctypes.byref(ex), 4, 0, p, 1, None, p, p, None, None, 1, 4,
                                   _lib.POE_VARIANT['A-noprior'], None) == -1


def test_binding_arity_matches_header_prototypes():
    """Every ctypes signature of _lib.py has exactly as many arguments as the prototype in include/mvae_hip.h (a
    ctypes call with one argument too few does not fail -- it passes garbage)."""
    text = open(os.path.join(ROOT, 'include', 'mvae_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(?:int|size_t|void|const char \*)\s*\**\s*(mvae_[a-zA-Z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ('', 'void') else args.count(',') + 1
    table = dict(_lib._SIGNATURES)
    table.update(_lib._TUNING_SIGNATURES)
    assert set(table) <= set(protos), sorted(set(table) - set(protos))
    wrong = ['%s: header %d, binding %d' % (n, protos[n], len(a)) for n, (_, a) in table.items() if protos[n] != len(a)]
    assert not wrong, '; '.join(wrong)

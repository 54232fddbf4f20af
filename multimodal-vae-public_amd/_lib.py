"""ctypes binding of libmvae_hip.so (the C ABI declared in include/mvae_hip.h).

The library is the product: there is no CPU / eager-PyTorch fallback.  ``lib()`` raises
``RuntimeError`` when the shared object has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C multimodal-vae-public_amd/csrc``), and every
wrapper in ``ops.py`` raises when it is handed a non-GPU tensor.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmvae_hip.so')

MVAE_OK = 0
ERRORS = {-1: 'MVAE_ERR_ARG (bad shape / null pointer / unsupported stride)',
          -2: 'MVAE_ERR_LAUNCH (hip kernel launch failed)',
          -3: 'MVAE_ERR_WS (workspace too small)',
          -4: 'MVAE_ERR_COMM (RCCL not loadable, or an RCCL / HIP runtime call of mvae_comm_* failed)'}
COMM_ID_BYTES = 128   # MVAE_COMM_ID_BYTES
ACT_SWISH = 1
ACCUMULATE = 2
POE_NO_PRIOR = 4     # MVAE_POE_NO_PRIOR
POE_VARIANT = {'A': 0, 'B': 1, 'A-noprior': 0 | POE_NO_PRIOR, 'B-noprior': 1 | POE_NO_PRIOR}
MAX_EXPERTS = 32
STATS_TILE_ELEMS = 512   # MVAE_STATS_TILE_ELEMS
MAX_TERMS = 40


class Experts(ctypes.Structure):
    _fields_ = [('mu', c_void_p * MAX_EXPERTS), ('logvar', c_void_p * MAX_EXPERTS)]


class ElboPart(ctypes.Structure):
    _fields_ = [('rows', c_void_p), ('coef', c_void_p), ('term_of', c_void_p), ('first_term', c_int),
                ('groups', c_int), ('rows_per_group', c_int)]


ELBO_MAX_PARTS = 4


class WgradItem(ctypes.Structure):          # mvae_wgrad_item
    _fields_ = [('dy', c_void_p), ('lddy', c_int), ('x', c_void_p), ('ldx', c_int), ('dw', c_void_p),
                ('db', c_void_p), ('M', c_int), ('N', c_int), ('K', c_int), ('flags', c_int)]


WGRAD_BATCH_MAX = 16


class AdamFuse(ctypes.Structure):           # mvae_adam_fuse
    _fields_ = [('grad_base', c_void_p), ('param_base', c_void_p), ('exp_avg_base', c_void_p),
                ('exp_avg_sq_base', c_void_p), ('coef2', c_void_p), ('beta1', ctypes.c_double),
                ('beta2', ctypes.c_double), ('eps', ctypes.c_double), ('grad_scale', ctypes.c_float)]


class RepackItem(ctypes.Structure):         # mvae_repack_item
    _fields_ = [('w', c_void_p), ('wr', c_void_p), ('transposed', c_int), ('Cin', c_int), ('Cout', c_int),
                ('stride', c_int), ('pad', c_int)]


REPACK_MAX = 16


class ExpertGrads(ctypes.Structure):
    _fields_ = [('dmu', c_void_p * MAX_EXPERTS), ('dlogvar', c_void_p * MAX_EXPERTS)]


P = c_void_p
_SIGNATURES = {
    'mvae_abi_version': (c_int, []),
    'mvae_gemm_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mvae_linear_fwd': (c_int, [P, c_int, P, P, P, P, c_int, P, c_float, c_int, c_int, c_int, P, c_size_t, P]),
    'mvae_linear_bce_fwd': (c_int, [P, c_int, P, P, P, c_int, c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, P]),
    'mvae_linear_ce_fwd': (c_int, [P, c_int, P, P, P, c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, P]),
    'mvae_linear_dgrad': (c_int, [P, c_int, P, P, c_int, P, P, c_float, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'mvae_linear_wgrad': (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'mvae_linear_fwd_grouped': (c_int, [P, c_int, c_size_t, P, c_size_t, P, c_size_t, P, P, c_int, c_size_t,
                                        c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'mvae_linear_dgrad_grouped': (c_int, [P, c_int, c_size_t, P, c_size_t, P, c_int, c_size_t, P, c_size_t,
                                          c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'mvae_linear_wgrad_grouped': (c_int, [P, c_int, c_size_t, P, c_int, c_size_t, P, c_size_t, P, c_size_t,
                                          c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'mvae_conv2d_k4_fwd': (c_int, [P, P, P, P] + [c_int] * 7 + [P]),
    'mvae_conv2d_k4_dgrad': (c_int, [P, P, P, P] + [c_int] * 7 + [P, c_size_t, P]),
    'mvae_conv2d_k4_wgrad': (c_int, [P, P, P] + [c_int] * 8 + [P, c_size_t, P]),
    'mvae_convT2d_k4_fwd': (c_int, [P, P, P, P] + [c_int] * 7 + [P, c_size_t, P]),
    'mvae_convT2d_k4_dgrad': (c_int, [P, P, P, P] + [c_int] * 7 + [P]),
    'mvae_convT2d_k4_wgrad': (c_int, [P, P, P] + [c_int] * 8 + [P, c_size_t, P]),
    'mvae_bn_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mvae_bn_train_fwd': (c_int, [P] * 8 + [c_int] * 4 + [c_float, c_float, c_int, P, c_int, P, c_size_t, P]),
    'mvae_bn_train_bwd': (c_int, [P] * 9 + [c_int] * 5 + [P, c_size_t, P]),
    'mvae_bn_eval_fwd': (c_int, [P] * 6 + [c_int] * 3 + [c_float, c_int, P]),
    'mvae_swish_fwd': (c_int, [P, P, c_size_t, P]),
    'mvae_swish_bwd': (c_int, [P, P, P, c_size_t, P]),
    'mvae_sigmoid_fwd': (c_int, [P, P, c_size_t, P]),
    'mvae_affine_fwd': (c_int, [P, P, P, P, c_size_t, c_size_t, P]),
    'mvae_embedding_swish_fwd': (c_int, [P, c_int, P, P, c_int, c_int, c_int, P]),
    'mvae_embedding_swish_bwd': (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    'mvae_embedding_swish_fwd_grouped': (c_int, [P, c_int, c_size_t, P, c_size_t, P, c_size_t, c_int, c_int, c_int,
                                                 c_int, P]),
    'mvae_embedding_swish_bwd_grouped': (c_int, [P, c_int, c_size_t, P, c_size_t, P, c_size_t, P, c_int, c_int,
                                                 c_int, c_int, c_int, P]),
    'mvae_resample_ksize': (c_int, [c_int, c_int]),
    'mvae_resample_coeffs': (c_int, [c_int, c_int, P, P]),
    'mvae_resize_crop_u8_to_f32': (c_int, [P, P] + [c_int] * 8 + [P, P, c_int, P, P, c_int, c_int, c_int, P]),
    'mvae_u8_to_f32': (c_int, [P, P, c_size_t, P]),
    'mvae_poe_fwd': (c_int, [ctypes.POINTER(Experts), c_int, c_int, P, c_int, P, P, P, P, P,
                             c_int, c_int, c_int, P]),
    'mvae_poe_fwd_draw': (c_int, [ctypes.POINTER(Experts), c_int, c_int, P, c_int, P, c_uint64, P, c_uint64, P, P, P, P,
                                  c_int, c_int, c_int, P]),
    'mvae_poe_bwd': (c_int, [ctypes.POINTER(Experts), c_int, c_int, P, c_int, P, P, P, P, P, P, P, c_int,
                             ctypes.POINTER(ExpertGrads), c_int, c_int, c_int, c_int, P]),
    'mvae_poe_bwd_split': (c_int, [ctypes.POINTER(Experts), c_int, c_int, P, c_int, P, P, P, P,
                                   ctypes.POINTER(c_int), P, ctypes.POINTER(c_int), P, c_int,
                                   ctypes.POINTER(ExpertGrads), c_int, c_int, c_int, c_int, P]),
    'mvae_kl_rows_fwd': (c_int, [P, P, P, c_int, c_int, P]),
    'mvae_kl_rows_bwd': (c_int, [P, P, P, P, P, c_int, c_int, P]),
    'mvae_bce_rowsum_fwd': (c_int, [P, P, P, P, P, P] + [c_int] * 7 + [P]),
    'mvae_bce_rowsum_bwd': (c_int, [P, P, P, P, P] + [c_int] * 7 + [P]),
    'mvae_ce_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'mvae_ce_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'mvae_group_sums': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'mvae_conv_k4_repack_floats': (c_size_t, [c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    'mvae_conv_k4_repack_batched': (c_int, [ctypes.POINTER(RepackItem), c_int, P]),
    'mvae_convT2d_k4_stats_tiles': (c_size_t, [c_int] * 7),
    'mvae_convT2d_k4_fwd_stats': (c_int, [P, P, P, c_size_t] + [c_int] * 7 + [P, c_size_t, P]),
    'mvae_bn_stats_merge': (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_float, c_float, c_int, P, P]),
    'mvae_linear_wgrad_batched': (c_int, [ctypes.POINTER(WgradItem), c_int, P]),
    'mvae_linear_wgrad_batched_adam': (c_int, [ctypes.POINTER(WgradItem), c_int, ctypes.POINTER(AdamFuse), P]),
    'mvae_adam_prepare': (c_int, [P, ctypes.c_int64, c_double, c_double, c_double, P, P]),
    'mvae_elbo_reduce': (c_int, [ctypes.POINTER(ElboPart), c_int, P, c_int, P, c_size_t, P, c_uint64, P]),
    'mvae_philox_fill': (c_int, [P, c_size_t, c_int, c_float, c_uint64, P, c_uint64, P]),
    'mvae_randn': (c_int, [P, c_size_t, c_uint64, P, P]),
    'mvae_bernoulli': (c_int, [P, c_size_t, c_float, c_uint64, P, P]),
    'mvae_adam_step': (c_int, [P, P, P, P, c_size_t, c_double, c_double, c_double, c_double, c_float, P, P]),
    'mvae_adam_apply': (c_int, [P, P, P, P, c_size_t, c_double, c_double, c_double, c_double, c_float, P, P]),
    'mvae_adam_apply_at': (c_int, [P, P, P, P, c_size_t, c_double, c_double, c_double, c_double, c_float, P,
                                   ctypes.c_int64, P]),
    'mvae_adam_apply_coef': (c_int, [P, P, P, P, c_size_t, P, c_double, c_double, c_double, c_float, P]),
    'mvae_counter_add': (c_int, [P, ctypes.c_int64, P]),
    'mvae_trace_marker': (c_int, [c_int, P]),
    'mvae_fill': (c_int, [P, c_size_t, c_float, P]),
    'mvae_ingest': (c_int, [P, P, c_size_t, P, P, c_size_t, P, P, c_size_t, P]),
    'mvae_reparam_fwd': (c_int, [P, P, P, P, c_size_t, P]),
    'mvae_reparam_bwd': (c_int, [P, P, P, P, P, c_size_t, P]),
    'mvae_dropout_fanout_fwd': (c_int, [P, P, P, c_float, c_int, c_int, c_int, P]),
    'mvae_dropout_fanin_bwd': (c_int, [P, P, P, c_float, c_int, c_int, c_int, P]),
    'mvae_block_gather': (c_int, [P, P, P, c_int, c_size_t, P]),
    'mvae_block_scatter_add': (c_int, [P, P, P, c_int, c_int, c_size_t, P]),
    'mvae_scatter_sums': (c_int, [P, P, P, P, P, c_int, c_int, P]),
    'mvae_bce_elem_fwd': (c_int, [P, P, P, c_size_t, P]),
    'mvae_bce_elem_bwd': (c_int, [P, P, P, P, P, c_size_t, P]),
    # K16: the recurrent text stacks of MultiMNIST
    'mvae_gru_cell_fwd': (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, P]),
    'mvae_gru_cell_bwd': (c_int, [P, c_int, P, c_int, P, P, c_int, P, P, P, c_int, c_int, P]),
    'mvae_embedding_fwd': (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'mvae_embedding_bwd': (c_int, [P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    'mvae_copy2d': (c_int, [P, c_int, P, c_int, P, c_int, c_float, c_int, c_int, c_int, P]),
    'mvae_argmax_rows': (c_int, [P, c_int, P, c_int, c_int, P]),
    # C1: the gradient exchange (RCCL bound at run time)
    'mvae_comm_use_library': (c_int, [ctypes.c_char_p]),
    'mvae_comm_rccl_version': (c_int, []),
    'mvae_comm_unique_id': (c_int, [P, c_size_t]),
    'mvae_comm_init': (c_int, [ctypes.POINTER(P), P, c_size_t, c_int, c_int, c_int]),
    'mvae_comm_rank': (c_int, [P]),
    'mvae_comm_world': (c_int, [P]),
    'mvae_comm_last_error': (ctypes.c_char_p, [P]),
    'mvae_comm_broadcast': (c_int, [P, P, c_size_t, c_int, P]),
    'mvae_comm_allreduce_async': (c_int, [P, P, c_size_t, P, ctypes.POINTER(c_int)]),
    'mvae_comm_wait': (c_int, [P, c_int, P]),
    'mvae_comm_async_error': (c_int, [P]),
    'mvae_comm_synchronize': (c_int, [P, P, c_int]),
    'mvae_comm_destroy': (c_int, [P]),
}

# tuning overrides: only exported by libmvae_hip_tuning.so (csrc built with -DMVAE_TUNING); bound when
# MVAE_HIP_LIB points the loader at that build (tools/gemm_bench.py, bench.py --force-tiling)
_TUNING_SIGNATURES = {
    'mvae_debug_set_tiling': (None, [c_int, c_int, c_int]),
    'mvae_debug_set_kwaves': (None, [c_int]),
    'mvae_debug_set_small': (None, [c_int, c_int]),
    'mvae_debug_set_split_target': (None, [ctypes.c_long]),
    'mvae_debug_set_knockout': (None, [c_int]),
}
TUNING_LIB_PATH = os.path.join(_HERE, 'libmvae_hip_tuning.so')

_lib = None


def exported_symbols():
    """Names include/mvae_hip.h declares (the not-gpu tests check the .so exports each)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get('MVAE_HIP_LIB') or LIB_PATH     # an alternative BUILD of the same library
        if not os.path.exists(path):
            raise RuntimeError(
                'libmvae_hip.so is not built (%s missing): run `make -C %s/csrc`; there is no '
                'CPU fallback for the MVAE HIP path.' % (path, _HERE))
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in _TUNING_SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        if handle.mvae_abi_version() != 6:
            raise RuntimeError('libmvae_hip.so ABI version mismatch')
        _lib = handle
    return _lib


def check(rc, what):
    if rc != MVAE_OK:
        raise RuntimeError('%s failed: %s' % (what, ERRORS.get(rc, 'error %d' % rc)))

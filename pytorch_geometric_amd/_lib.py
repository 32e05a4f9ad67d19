"""ctypes binding of ``libpyg_amd.so`` — the C ABI declared in ``include/pyg_amd.h``.

This module is the only place that touches the shared object.  There is deliberately no CPU
fallback: if the library is missing (and cannot be built) or a call fails, we raise.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_uint64, c_void_p)

from . import _build

ABI_VERSION = 10
X_DENSE, X_COMPRESSED = 0, 1  # pygamd_x_format  # PYGAMD_ABI_VERSION of include/pyg_amd.h
IDX_I32, IDX_I64 = 0, 1
SUM, MEAN, MIN, MAX, MUL, ANY = 0, 1, 2, 3, 4, 5
REDUCE_IDS = {'sum': SUM, 'add': SUM, 'mean': MEAN, 'min': MIN, 'amin': MIN, 'max': MAX,
              'amax': MAX, 'mul': MUL, 'any': ANY}


class PygAmdError(RuntimeError):
    pass


class SpmmArgs(Structure):
    """Mirror of ``pygamd_spmm_args`` (include/pyg_amd.h)."""
    _fields_ = [
        ('rowptr', c_void_p), ('col', c_void_p), ('eid', c_void_p), ('w', c_void_p),
        ('src_scale', c_void_p), ('x', c_void_p), ('out', c_void_p), ('arg_out', c_void_p),
        ('n_rows', c_int64), ('n_src', c_int64), ('F', c_int64), ('ldx', c_int64),
        ('ldo', c_int64), ('idx_dtype', c_int32), ('reduce', c_int32), ('w_heads', c_int32),
        ('head_dim', c_int32), ('hub_rows', c_void_p), ('hub_chunk_ptr', c_void_p),
        ('n_hub', c_int64), ('n_chunks', c_int64), ('hub_threshold', c_int64),
        ('hub_chunk', c_int64), ('accumulate', c_int32), ('hub_phase', c_int32),
        ('arg32_out', c_void_p), ('relu_mask', c_void_p), ('ld_mask', c_int64),
        ('relu_bits', c_void_p), ('ld_bits', c_int64), ('src_bits', c_void_p),
        ('src_bits_set', c_void_p), ('x_format', c_int32), ('reserved0', c_int32),
        ('rowend', c_void_p), ('accumulate_rows', c_int64),
    ]


class SageFusedArgs(Structure):
    """Mirror of ``pygamd_sage_fused_args`` (include/pyg_amd.h)."""
    _fields_ = [
        ('x_root', c_void_p), ('ld_root', c_int64), ('w', c_void_p), ('ldw', c_int64),
        ('bias', c_void_p), ('Fo', c_int64), ('relu', c_int32), ('save_agg', c_int32),
        ('y', c_void_p), ('ldy', c_int64), ('relu_bits_out', c_void_p), ('ld_bits_out', c_int64),
        ('mask_bits', c_void_p), ('ld_mask_bits', c_int64), ('row_scale', c_void_p),
        ('y_scaled', c_void_p), ('ldy_scaled', c_int64),
        ('compressed_out', c_void_p), ('ld_compressed', c_int64),
    ]


# name -> (restype, argtypes); must list every PYGAMD_API symbol of include/pyg_amd.h
_P = c_void_p
SIGNATURES = {
    'pygamd_abi_version': (c_int, []),
    'pygamd_status_string': (c_char_p, [c_int]),
    'pygamd_last_hip_error': (c_int, []),
    'pygamd_build_arch': (c_char_p, []),
    'pygamd_index_sort_workspace_bytes': (c_int, [c_int, c_int64, POINTER(c_size_t)]),
    'pygamd_index_sort': (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    'pygamd_index2ptr': (c_int, [_P, c_int, c_int64, c_int64, _P, _P]),
    'pygamd_ptr2index': (c_int, [_P, c_int, c_int64, c_int64, _P, _P]),
    'pygamd_index_minmax': (c_int, [_P, c_int, c_int64, _P, _P]),
    'pygamd_permute_index': (c_int, [_P, c_int, _P, c_int64, _P, _P]),
    'pygamd_cast_index': (c_int, [_P, c_int64, c_int, _P, _P]),
    'pygamd_index_guard': (c_int, [_P, c_int, c_int64, c_int64, _P, c_int, _P, _P]),
    'pygamd_cumsum_workspace_bytes': (c_int, [c_int, c_int64, POINTER(c_size_t)]),
    'pygamd_cumsum': (c_int, [_P, c_int, c_int64, _P, _P, c_size_t, _P]),
    'pygamd_hub_plan_workspace_bytes': (c_int, [c_int, c_int64, POINTER(c_size_t)]),
    'pygamd_hub_plan': (c_int, [_P, c_int, c_int64, c_int64, c_int64, _P, _P, c_int64,
                                POINTER(c_int64), POINTER(c_int64), _P, c_size_t, _P]),
    'pygamd_rows_compress': (c_int, [_P, c_int64, c_int64, c_int64, _P, c_int64, _P]),
    'pygamd_rows_pack': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, c_int64, c_int64, _P,
                                 c_int64, c_int64, _P, _P, _P]),
    'pygamd_spmm_csr_workspace_bytes': (c_int, [POINTER(SpmmArgs), POINTER(c_size_t)]),
    'pygamd_spmm_csr': (c_int, [POINTER(SpmmArgs), _P, c_size_t, _P]),
    'pygamd_spmm_csr_tie_count': (c_int, [_P, _P, c_int, _P, c_int64, _P, c_int64, c_int64,
                                          c_int64, c_int, _P, _P]),
    'pygamd_spmm_csr_minmax_backward': (c_int, [_P, _P, c_int, _P, c_int64, _P, _P, _P, c_int64,
                                                c_int64, c_int64, _P, c_int64, _P]),
    'pygamd_spmm_csr_minmax_backward_dst': (c_int, [_P, _P, c_int, _P, c_int64, _P, c_int64, _P,
                                                    c_int64, c_int64, c_int64, c_int64, c_int, _P,
                                                    c_int64, _P]),
    'pygamd_multi_reduce_csr': (c_int, [_P, _P, c_int, _P, c_int64, c_int64, c_int64, _P, _P, _P,
                                        _P, c_int64, _P]),
    'pygamd_minmax_backward_src_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int64]),
    'pygamd_spmm_csr_minmax_backward_src': (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int64, _P,
                                                    c_int64, _P, c_int64, c_int64, c_int64,
                                                    c_int64, c_int64, c_int, _P, c_size_t, _P,
                                                    c_int64, _P]),
    'pygamd_spmm_csr_minmax_backward_arg': (c_int, [_P, _P, c_int, _P, _P, c_int64, _P, c_int64, _P,
                                                    c_int64, c_int64, c_int64, c_int64, c_int, _P,
                                                    c_int64, _P]),
    'pygamd_sddmm_csr': (c_int, [_P, _P, _P, c_int, _P, c_int64, _P, c_int64, c_int64, c_int64,
                                 c_int32, c_int32, _P, _P]),
    'pygamd_sddmm_spmm_csr': (c_int, [_P, _P, _P, c_int, _P, c_int64, _P, c_int64, c_int64,
                                      c_int64, c_int32, c_int32, _P, _P, _P, c_int64, _P]),
    'pygamd_colsum': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    'pygamd_bias_act': (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int, _P, c_int64, _P]),
    'pygamd_relu_backward_colsum': (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int64, _P,
                                            c_int64, _P, _P]),
    'pygamd_sage_layer_forward_supported': (c_int, [c_int64, c_int64, c_int]),
    'pygamd_sage_layer_forward': (c_int, [POINTER(SpmmArgs), _P, c_int64, _P, c_int64, _P, c_int64,
                                          c_int, c_int, _P, c_int64, _P, c_int64, _P, c_size_t,
                                          _P]),
    'pygamd_sage_layer_fused': (c_int, [POINTER(SpmmArgs), POINTER(SageFusedArgs), _P, c_size_t,
                                        _P]),
    'pygamd_sage_layer_fused_workspace_bytes': (c_int, [POINTER(SpmmArgs), POINTER(SageFusedArgs),
                                                        POINTER(c_size_t)]),
    'pygamd_linear_nt_workspace_bytes': (c_int, [c_int64, c_int64, c_int64, POINTER(c_size_t)]),
    'pygamd_linear_forward': (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64,
                                      c_int, c_int, _P, c_int64, _P, c_size_t, _P]),
    'pygamd_set_gemm_mode': (c_int, [c_int]),
    'pygamd_get_gemm_mode': (c_int, []),
    'pygamd_linear_dgrad': (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64,
                                    c_int64, c_int, _P, c_int64, _P, c_int64, _P, c_int64, _P]),
    'pygamd_linear_dgrad2': (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64,
                                     c_int64, c_int, _P, c_int64, _P, c_int64, _P, c_int64, _P,
                                     c_int64, _P, c_size_t, _P]),
    'pygamd_linear_wgrad_workspace_bytes': (c_int, [c_int64, c_int64, c_int64,
                                                    POINTER(c_size_t)]),
    'pygamd_linear_wgrad': (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, c_int,
                                    c_int, _P, c_int64, _P, _P, c_size_t, _P]),
    'pygamd_linear_wgrad2': (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                     c_int64, c_int64, c_int, c_int, _P, c_int64, _P, _P,
                                     c_size_t, _P]),
    'pygamd_segment_matmul_tile_rows': (c_int, []),
    'pygamd_segment_matmul_workspace_bytes': (c_int, [c_int64, c_int64, c_int64,
                                                      POINTER(c_size_t)]),
    'pygamd_segment_matmul': (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int64,
                                      _P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P,
                                      c_size_t, _P]),
    'pygamd_segment_matmul_wgrad': (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64,
                                            c_int64, c_int64, c_int64, _P, _P]),
    'pygamd_sample_max_fanout': (c_int, []),
    'pygamd_sample_neighbors': (c_int, [_P, _P, c_int, _P, c_int64, _P, c_int64, c_uint64, c_int,
                                        _P, _P, _P, _P, _P]),
    'pygamd_slots_max_fanout': (c_int, []),
    'pygamd_slots_max_hops': (c_int, []),
    'pygamd_slots_seed': (c_int, [_P, c_int, c_int64, _P, _P, _P, _P]),
    'pygamd_slots_sample': (c_int, [_P, _P, c_int, _P, c_int64, c_int64, c_int, c_int64, c_int64,
                                    c_uint64, c_int, _P, _P, _P, _P, _P, _P]),
    'pygamd_slots_resolve': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_int,
                                     _P]),
    'pygamd_slots_gather': (c_int, [_P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P]),
    'pygamd_slots_transpose': (c_int, [_P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    'pygamd_cross_entropy_step_workspace_bytes': (c_int, [c_int64, _P]),
    'pygamd_cross_entropy_step': (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P,
                                          c_int64, _P, _P, c_size_t, _P, _P, _P]),
    'pygamd_adam_step': (c_int, [_P, _P, _P, _P, c_int64, _P, c_int64, c_double, c_double,
                                 c_double, c_double, c_double, c_double, _P, c_int, _P, _P, _P,
                                 _P, _P]),
    'pygamd_edge_key': (c_int, [_P, _P, c_int, c_int64, c_int64, c_int, _P, _P]),
    'pygamd_run_flags': (c_int, [_P, c_int64, _P, _P]),
    'pygamd_edge_unkey': (c_int, [_P, _P, _P, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P]),
    'pygamd_sample_counts': (c_int, [_P, c_int, _P, c_int64, c_int64, c_int, _P, _P, _P]),
    'pygamd_relabel': (c_int, [c_int, _P, c_int, c_int64, _P, _P, _P, c_int64, _P, _P, _P]),
    'pygamd_gather_rows': (c_int, [_P, c_int64, c_int64, _P, c_int, c_int64, c_int64, _P,
                                   c_int64, _P, _P]),
    'pygamd_gather_scatter_add': (c_int, [_P, c_int64, _P, _P, c_int, _P, _P, c_int64, _P,
                                          c_int64, _P, c_int64, _P]),
    'pygamd_scatter_init': (c_int, [_P, c_int64, c_int64, c_int64, c_int, _P, _P]),
    'pygamd_scatter_rows': (c_int, [_P, c_int64, _P, c_int, c_int64, c_int64, _P, c_int64,
                                    c_int64, c_int, _P, _P, _P]),
    'pygamd_scatter_finalize': (c_int, [_P, c_int64, c_int64, c_int64, c_int, _P, _P]),
    'pygamd_scatter_minmax_tie_count': (c_int, [_P, c_int64, _P, c_int, c_int64, c_int64, _P,
                                                c_int64, c_int64, _P, _P]),
    'pygamd_scatter_minmax_backward': (c_int, [_P, c_int64, _P, c_int, c_int64, c_int64, _P, _P,
                                               _P, c_int64, c_int64, _P, c_int64, _P]),
    'pygamd_scatter_mul_backward_workspace_bytes': (c_int, [c_int64, c_int64, POINTER(c_size_t)]),
    'pygamd_scatter_mul_backward': (c_int, [_P, c_int64, _P, c_int, c_int64, c_int64, _P, _P,
                                            c_int64, c_int64, _P, c_int64, _P, c_size_t, _P]),
    'pygamd_scatter_argmax': (c_int, [_P, _P, c_int, c_int64, c_int64, _P, _P, _P]),
    'pygamd_softmax_index_forward': (c_int, [_P, _P, c_int, c_int64, c_int64, c_int64, _P, _P,
                                             _P, _P]),
    'pygamd_softmax_index_backward': (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_int64, _P,
                                              _P, _P]),
    'pygamd_segment_softmax_forward': (c_int, [_P, _P, c_int, c_int64, c_int64, _P, _P]),
    'pygamd_segment_softmax_backward': (c_int, [_P, _P, _P, c_int, c_int64, c_int64, _P, _P]),
    'pygamd_segment_logsumexp_forward': (c_int, [_P, _P, c_int, c_int64, c_int64, _P, _P]),
    'pygamd_segment_logsumexp_backward': (c_int, [_P, _P, _P, _P, c_int, c_int64, c_int64, _P,
                                                  _P]),
    'pygamd_head_dot_forward': (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_int64, _P, _P,
                                        _P]),
    'pygamd_head_dot_backward': (c_int, [_P, c_int64, _P, _P, _P, _P, c_int64, c_int64, c_int64,
                                         _P, c_int64, c_int, _P, _P, _P]),
    'pygamd_gat_edge_softmax_forward': (c_int, [_P, _P, c_int, _P, _P, c_int64, c_int64,
                                                c_float, _P, _P]),
    'pygamd_gat_edge_softmax_backward': (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int64,
                                                 c_int64, c_float, _P, _P, _P]),
}

# include/pyg_amd_lab.h: schedules measured and not adopted + timing probes (NOT the boundary;
# exported by libpyg_amd_lab.so only)
LAB_SIGNATURES = {
    'pygamd_lab_sage_layer_fused': (c_int, [POINTER(SpmmArgs), POINTER(SageFusedArgs), c_int,
                                            c_int, _P, c_size_t, _P]),
    'pygamd_lab_set_wgrad_variant': (c_int, [c_int]),
    'pygamd_lab_copy': (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
}

_lib = None       # libpyg_amd.so: the product library (include/pyg_amd.h, nothing else)
_lab = None       # libpyg_amd_lab.so: the same sources + the laboratory entry points
_use_lab = False  # route load() to the laboratory build (tests / scripts that select a variant)


def lib_path():
    return _build.LIB_PATH


GEMM_MODES = {'fp32': 0, 'split': 1}  # PYGAMD_GEMM_FP32 / PYGAMD_GEMM_SPLIT_BF16


def _open(path, signatures):
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in signatures:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pygamd_abi_version() != ABI_VERSION:
        raise PygAmdError(f'ABI mismatch: {path} reports {lib.pygamd_abi_version()}')
    return lib


def _default_mode():
    # 'split' (round 4): error against fp64 at or below the fp32 matrix instruction's on the same
    # inputs (tests/test_gpu_split_accept.py, test_gpu_gemm.py), 2.7 x fewer matrix-pipe cycles;
    # PYGAMD_GEMM_MODE=fp32 selects the exact instruction (bitwise an fmaf chain)
    mode = os.environ.get('PYGAMD_GEMM_MODE', 'split')
    if mode not in GEMM_MODES:
        raise PygAmdError(f"PYGAMD_GEMM_MODE must be one of {sorted(GEMM_MODES)}, got '{mode}'")
    return GEMM_MODES[mode]


def load():
    """The product library (building first if the in-tree .so is missing/stale and hipcc
    exists) — or the laboratory build while :func:`use_lab` routes there."""
    global _lib
    if _use_lab:
        return load_lab()
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if _build.is_stale():
        if _build.find_hipcc() is not None and os.environ.get('PYG_AMD_NO_BUILD') != '1':
            _build.build_library(verbose=False)
        elif not os.path.exists(path):
            raise PygAmdError(
                f"{path} is missing and hipcc is unavailable: run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` on a machine with ROCm. "
                f"There is no CPU fallback for the pytorch_geometric_amd kernels.")
    lib = _open(path, list(SIGNATURES.items()))
    lib.pygamd_set_gemm_mode(_default_mode())
    _lib = lib
    return lib


def load_lab():
    """``libpyg_amd_lab.so``: the product sources plus csrc/sage_fused_lab.hip and the weight-
    gradient probes (include/pyg_amd_lab.h) — schedules measured and not adopted, timing probes,
    the copy-rate probe of bench.py's side figure.  Loaded by scripts/, the tests that pin those
    schedules to the production results, and that side figure; never by the product path."""
    global _lab
    if _lab is not None:
        return _lab
    path = _build.LAB_LIB_PATH
    if _build.lab_is_stale():
        if _build.find_hipcc() is not None and os.environ.get('PYG_AMD_NO_BUILD') != '1':
            _build.build_lab_library(verbose=False)
        elif not os.path.exists(path):
            raise PygAmdError(f"{path} is missing and hipcc is unavailable (the laboratory "
                              f"build: `python -m pytorch_geometric_amd._build`)")
    lab = _open(path, list(SIGNATURES.items()) + list(LAB_SIGNATURES.items()))
    # (its own copy of the process-wide switches: start from the product library's mode)
    lab.pygamd_set_gemm_mode(_lib.pygamd_get_gemm_mode() if _lib is not None else _default_mode())
    _lab = lab
    return lab


def use_lab(flag: bool) -> None:
    global _use_lab
    _use_lab = bool(flag)
    if _use_lab:
        load_lab()


def lab_active() -> bool:
    return _use_lab


def loaded():
    """The library objects that are open (for process-wide switches kept per library)."""
    if _lib is None and _lab is None:
        load()
    return [l for l in (_lib, _lab) if l is not None]


def check(rc, what=''):
    if rc != 0:
        lib = load()
        msg = lib.pygamd_status_string(rc).decode()
        extra = f' (hipError {lib.pygamd_last_hip_error()})' if rc == 4 else ''
        raise PygAmdError(f'{what or "pygamd call"} failed: {msg}{extra}')

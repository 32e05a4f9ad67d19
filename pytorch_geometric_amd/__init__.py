"""pytorch_geometric_amd — MI355X (gfx950) native message-passing / aggregation backend.

The hot path of PyG's ``MessagePassing.propagate`` (gather on ``edge_index[0]``, edge-wise message,
scatter-{add, mean, max} onto ``edge_index[1]``) and the CSR SpMM path of SAGEConv / GCNConv /
GATConv, as hand-written HIP kernels behind a C ABI (``include/pyg_amd.h`` ->
``lib/libpyg_amd.so``).  The Python layer mirrors the reference's operator interface
(``utils.scatter / segment / softmax / spmm / index_sort``, ``nn.SAGEConv`` ...) and
``backend.install()`` rebinds the same names inside an importable ``torch_geometric``.

There is no CPU fallback: every op needs HIP device tensors and the built library.
"""
from . import _build
from ._lib import PygAmdError, lib_path, load as load_library
from .edge_index import EdgeIndex, as_edge_index, clear_cache, set_cache_enabled
from .index import index2ptr, ptr2index
from . import utils
from . import nn

__version__ = '0.1.0'


def set_gemm_mode(mode: str) -> str:
    """Arithmetic of the dense feature-transform kernels (csrc/gemm.hip): ``'fp32'`` (default —
    the fp32 matrix instruction, bitwise an fmaf chain) or ``'split'`` (every fp32 operand as three
    bf16 terms, six bf16 matrix products, fp32 accumulation: same fp32 tensors in and out, error
    against fp64 at or below the fmaf chain's).  Also ``PYGAMD_GEMM_MODE``.  Returns the previous
    mode."""
    from . import _native
    return _native.set_gemm_mode(mode)


def get_gemm_mode() -> str:
    from . import _native
    return _native.get_gemm_mode()


def check_index_errors() -> None:
    """Scatter kernels skip indices outside ``[0, dim_size)`` and flag them; the flag reaches the
    host asynchronously (``PYGAMD_CHECK_INDEX``: ``async`` default | ``sync`` | ``off``).  This
    waits for every scatter launched so far and raises ``IndexError`` if one was flagged — the
    counterpart of synchronising after a device-side assert in the reference's GPU path."""
    from . import _native
    _native.check_index_errors()


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 into ``lib/libpyg_amd.so``, the compiled PyTorch binding
    (csrc/torch_binding.cpp) into ``lib/libpyg_amd_torch.so`` and the laboratory build
    (``lib/libpyg_amd_lab.so``: scripts / tests / bench side figure only) — all in-tree."""
    path = _build.build_library(force=force)
    _build.build_torch_binding(force=force)
    _build.build_lab_library(force=force)
    return path


def binding_status() -> str:
    """'compiled (<path>)' when ``torch.ops.pyg_amd_c`` (the TORCH_LIBRARY binding) carries the hot
    entry points, 'ctypes (...)' when the ctypes route to the same C ABI is in use."""
    from . import _compiled
    return _compiled.status()


__all__ = ['EdgeIndex', 'as_edge_index', 'clear_cache', 'set_cache_enabled', 'index2ptr',
           'ptr2index', 'utils', 'nn', 'build', 'load_library', 'lib_path', 'PygAmdError',
           'set_gemm_mode', 'get_gemm_mode', 'check_index_errors', 'binding_status']

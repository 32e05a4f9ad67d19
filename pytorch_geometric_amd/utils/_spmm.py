import weakref
from typing import Optional, Union

import torch
from torch import Tensor

from .._functions import SpmmFunction
from ..edge_index import EdgeIndex
from ._scatter import _require_fp32

_sparse_cache = {}


def _handle_from_sparse(src: Tensor):
    """(handle, values in COO order of the handle | None) for a torch sparse ``adj_t`` whose rows
    are destinations (utils/_spmm.py:57-111).  CSR is adopted as is; COO / CSC are converted once
    and cached per tensor."""
    key = id(src)
    hit = _sparse_cache.get(key)
    if hit is not None and hit[0]() is src:
        return hit[1], hit[2]
    n_dst, n_src = src.size(0), src.size(1)
    if src.layout == torch.sparse_csr:
        handle = EdgeIndex.from_csr(src.crow_indices(), src.col_indices(), (n_src, n_dst))
        value = src.values()
    elif src.layout == torch.sparse_coo:
        src_c = src.coalesce()
        row, col = src_c.indices()[0], src_c.indices()[1]
        handle = EdgeIndex(torch.stack([col, row]), (n_src, n_dst), sort_order='col')
        value = src_c.values()
    elif src.layout == torch.sparse_csc:
        col = _ptr_to_index(src.ccol_indices(), src.row_indices().numel())
        handle = EdgeIndex(torch.stack([col, src.row_indices()]), (n_src, n_dst),
                           sort_order='row')
        value = src.values()
    else:
        raise ValueError(f"unsupported sparse layout '{src.layout}'")
    if len(_sparse_cache) >= 8:
        _sparse_cache.pop(next(iter(_sparse_cache)))
    _sparse_cache[key] = (weakref.ref(src), handle, value)
    return handle, value


def _ptr_to_index(ptr: Tensor, n: int) -> Tensor:
    from .. import _native
    return _native.ptr2index(ptr, n)


def spmm(src: Union[EdgeIndex, Tensor], other: Tensor, reduce: str = 'sum',
         value: Optional[Tensor] = None) -> Tensor:
    r"""Sparse-dense product ``out = A @ other`` with reduce in ``sum | mean | min | max`` — the
    role of ``torch_geometric.utils.spmm`` (torch_geometric/utils/_spmm.py:12-136).  ``src`` is
    an :class:`EdgeIndex` handle (``A[i, j] = value[e]`` for every edge ``e = (j -> i)``) or a
    ``torch.sparse`` CSR / COO / CSC tensor in ``adj_t`` orientation (rows = destinations), i.e.
    what ``message_and_aggregate`` receives."""
    reduce = 'sum' if reduce == 'add' else reduce
    if reduce not in ('sum', 'mean', 'min', 'max'):
        raise ValueError(f"`reduce` argument '{reduce}' not supported")
    _require_fp32(other, 'spmm')
    if isinstance(src, EdgeIndex):
        return SpmmFunction.apply(other, value, src, reduce, 'coo')
    if isinstance(src, Tensor) and src.layout in (torch.sparse_csr, torch.sparse_coo,
                                                  torch.sparse_csc):
        if src.dim() != 2:
            raise ValueError("'src' must be a two-dimensional sparse matrix")
        handle, vals = _handle_from_sparse(src)
        if vals.dim() != 1:
            raise ValueError("only scalar sparse values are supported")
        if reduce in ('min', 'max'):
            # torch.sparse.mm(A, B, 'amax') — the reference's CPU route — multiplies by the
            # stored values; the extremum kernels take no weights, so only unit values are
            # accepted (the reference's CUDA route refuses min/max altogether,
            # utils/_spmm.py:92-99)
            if not bool((vals == 1).all()):
                raise NotImplementedError(
                    f"`{reduce}` reduction over a sparse matrix with non-unit values is not "
                    f"supported on the MI355X path")
            return SpmmFunction.apply(other, None, handle, reduce, 'coo')
        return SpmmFunction.apply(other, vals.to(torch.float32), handle, reduce, 'coo')
    raise ValueError("'src' must be an EdgeIndex handle or a torch.sparse CSR/COO/CSC tensor")

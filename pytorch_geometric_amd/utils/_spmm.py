from typing import Optional

from torch import Tensor

from .._functions import SpmmFunction
from ..edge_index import EdgeIndex
from ._scatter import _require_fp32


def spmm(src: EdgeIndex, other: Tensor, reduce: str = 'sum',
         value: Optional[Tensor] = None) -> Tensor:
    r"""Sparse-dense product ``out = A @ other`` with ``A[i, j] = value[e]`` for every edge
    ``e = (j -> i)`` of the handle and reduce in ``sum | mean | min | max`` — the role of
    ``torch_geometric.utils.spmm`` (torch_geometric/utils/_spmm.py:12-136) with
    ``adj_t``-orientation (rows = destinations), i.e. what ``message_and_aggregate`` computes."""
    reduce = 'sum' if reduce == 'add' else reduce
    if reduce not in ('sum', 'mean', 'min', 'max'):
        raise ValueError(f"`reduce` argument '{reduce}' not supported")
    if not isinstance(src, EdgeIndex):
        raise ValueError("'src' must be a pytorch_geometric_amd.EdgeIndex handle")
    _require_fp32(other, 'spmm')
    return SpmmFunction.apply(other, value, src, reduce, 'coo')

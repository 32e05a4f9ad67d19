from typing import Optional

from torch import Tensor

from .. import _native
from .._functions import IndexSoftmaxFunction, SegmentSoftmaxFunction
from ._scatter import _require_fp32
from .num_nodes import maybe_num_nodes


def softmax(src: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
            num_nodes: Optional[int] = None, dim: int = 0) -> Tensor:
    r"""Sparsely evaluated softmax: groups the rows of :obj:`src` by :obj:`index` (or by the CSR
    pointer :obj:`ptr` for sorted inputs) and normalises within each group — drop-in for
    ``torch_geometric.utils.softmax`` (torch_geometric/utils/_softmax.py:12-92), including the
    detached maximum and the ``1e-16`` added to the denominator."""
    _require_fp32(src, 'softmax')
    dim = dim + src.dim() if dim < 0 else dim
    if ptr is not None and ptr.dim() == 1 and dim == 0:
        return SegmentSoftmaxFunction.apply(src, ptr)
    if ptr is not None and index is None:
        index = _native.ptr2index(ptr, src.size(dim))
    if index is None:
        raise NotImplementedError("'softmax' requires 'index' to be specified")
    # index branch (:82-88: scatter-max (detached) -> gather -> exp -> scatter-sum -> gather -> div)
    # as one operator: group maxima, exp + group sums, normalisation (csrc/softmax.hip)
    N = maybe_num_nodes(index, num_nodes)
    if dim != 0:
        return softmax(src.movedim(dim, 0).contiguous(), index, None, N, 0).movedim(0, dim)
    if index.dim() != 1 or index.numel() != src.size(0):
        raise ValueError(f"'index' must hold one group id per entry of dimension {dim} of 'src' "
                         f"({src.size(0)}), got {tuple(index.shape)}")
    return IndexSoftmaxFunction.apply(src, index, N)

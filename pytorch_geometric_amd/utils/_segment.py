from torch import Tensor

from .._functions import SegmentFunction, SegmentLogSumExpFunction
from ._scatter import _require_fp32


def segment(src: Tensor, ptr: Tensor, reduce: str = 'sum') -> Tensor:
    r"""Reduces the rows of :obj:`src` within the ranges given by the monotone pointer
    :obj:`ptr` (``ptr[0] = 0``, ``ptr[-1] = src.size(0)``) — drop-in for
    ``torch_geometric.utils.segment`` (torch_geometric/utils/_segment.py:11-50).  Empty segments
    give 0 for every reduce, as the reference does."""
    if ptr.dim() != 1:
        raise ImportError("'segment' in an arbitrary dimension requires the 'torch-scatter' "
                          "package")
    if reduce not in ('sum', 'add', 'mean', 'min', 'max'):
        raise ValueError(f"Encountered invalid `reduce` argument '{reduce}'")
    _require_fp32(src, 'segment')
    reduce = 'sum' if reduce == 'add' else reduce
    return SegmentFunction.apply(src, ptr, reduce)


def segment_logsumexp(src: Tensor, ptr: Tensor, dim: int) -> Tensor:
    r"""Log of the summed exponentials of the slices of :obj:`src` along :obj:`dim` that lie in the
    same :obj:`ptr` range — drop-in for ``torch_geometric.utils.segment_logsumexp``
    (torch_geometric/utils/_segment.py:53-80): evaluated with the segment maximum subtracted, an
    empty segment gives 0.  One HIP kernel forward, one backward (the in-segment softmax)."""
    if ptr.dim() != 1:
        raise ValueError("'ptr' must be one-dimensional")
    _require_fp32(src, 'segment_logsumexp')
    dim = dim + src.dim() if dim < 0 else dim
    if dim != 0:
        return segment_logsumexp(src.transpose(0, dim).contiguous(), ptr, 0).transpose(0, dim)
    return SegmentLogSumExpFunction.apply(src, ptr)

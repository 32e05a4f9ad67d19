from torch import Tensor

from .._functions import SegmentFunction
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

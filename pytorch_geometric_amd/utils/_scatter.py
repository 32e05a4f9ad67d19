from typing import Optional

import torch
from torch import Tensor

from .. import _native
from .._functions import ScatterFunction

_REDUCES = ('sum', 'add', 'mean', 'min', 'max', 'amin', 'amax', 'mul', 'any')


def _require_fp32(src: Tensor, what: str):
    if src.dtype != torch.float32:
        raise NotImplementedError(
            f"pytorch_geometric_amd.{what} computes in float32 only (got {src.dtype}); cast the "
            f"input or use the reference path for other dtypes")


def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None,
            reduce: str = 'sum') -> Tensor:
    r"""Reduces all values from :obj:`src` at the indices given by the one-dimensional
    :obj:`index` along dimension :obj:`dim` — drop-in for
    ``torch_geometric.utils.scatter`` (torch_geometric/utils/_scatter.py:14-138).

    Semantics kept from the reference's CPU path: empty groups give 0 (1 for ``"mul"``);
    ``"mean"`` divides by ``count.clamp(min=1)``; ``dim_size=None`` means ``index.max() + 1``
    (one host sync); min/max gradients are split evenly over ties, the zero-initialised output
    counting as one more tie when the group extremum equals 0.
    """
    if isinstance(index, Tensor) and index.dim() != 1:
        raise ValueError(f"The `index` argument must be one-dimensional "
                         f"(got {index.dim()} dimensions)")
    dim = src.dim() + dim if dim < 0 else dim
    if isinstance(src, Tensor) and (dim < 0 or dim >= src.dim()):
        raise ValueError(f"The `dim` argument must lay between 0 and "
                         f"{src.dim() - 1} (got {dim})")
    if reduce not in _REDUCES:
        raise ValueError(f"Encountered invalid `reduce` argument '{reduce}'")
    if dim_size is None:
        dim_size = _native.index_minmax(index)[1] + 1 if index.numel() > 0 else 0
    _require_fp32(src, 'scatter')
    if src.size(dim) != index.numel():
        raise ValueError(f"'src' has {src.size(dim)} entries along dim {dim} but 'index' has "
                         f"{index.numel()}")
    reduce = {'add': 'sum', 'amin': 'min', 'amax': 'max'}.get(reduce, reduce)
    if dim != 0:
        out = ScatterFunction.apply(src.movedim(dim, 0).contiguous(), index, dim_size, reduce)
        return out.movedim(0, dim)
    return ScatterFunction.apply(src, index, dim_size, reduce)


def scatter_argmax(src: Tensor, index: Tensor, dim: int = 0,
                   dim_size: Optional[int] = None) -> Tensor:
    r"""Arg-max per group for one-dimensional inputs (torch_geometric/utils/_scatter.py:147-184):
    the last position attaining the group maximum; ``dim_size - 1`` for empty groups."""
    assert src.dim() == 1 and index.dim() == 1
    assert dim == 0 or dim == -1
    assert src.numel() == index.numel()
    if dim_size is None:
        dim_size = _native.index_minmax(index)[1] + 1 if index.numel() > 0 else 0
    _require_fp32(src, 'scatter_argmax')
    return _native.scatter_argmax(src.detach(), index, dim_size)

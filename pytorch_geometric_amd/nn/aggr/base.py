from typing import Optional

import torch
from torch import Tensor

from ... import _native
from ...utils import scatter, segment


class Aggregation(torch.nn.Module):
    r"""Base class of the aggregation operators: ``forward(x, index, ptr, dim_size, dim)`` reduces
    the rows of ``x`` that share an ``index`` (or lie in the same ``ptr`` range).  Contract and
    validation follow torch_geometric/nn/aggr/base.py:101-185."""

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        raise NotImplementedError

    def reset_parameters(self):
        pass

    def __call__(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                 dim_size: Optional[int] = None, dim: int = -2, **kwargs) -> Tensor:
        if dim >= x.dim() or dim < -x.dim():
            raise ValueError(f"Encountered invalid dimension '{dim}' of "
                             f"source tensor with {x.dim()} dimensions")
        if index is None and ptr is None:
            index = x.new_zeros(x.size(dim), dtype=torch.long)
        if ptr is not None:
            if dim_size is None:
                dim_size = ptr.numel() - 1
            elif dim_size != ptr.numel() - 1:
                raise ValueError(f"Encountered invalid 'dim_size' (got "
                                 f"'{dim_size}' but expected "
                                 f"'{ptr.numel() - 1}')")
        given = index is not None and dim_size is not None
        if index is not None and dim_size is None:
            dim_size = _native.index_minmax(index)[1] + 1 if index.numel() > 0 else 0
        prev = _native.set_error_style('dim_size')
        try:
            out = super().__call__(x, index=index, ptr=ptr, dim_size=dim_size, dim=dim, **kwargs)
            # a caller-supplied `dim_size` is the one way an index can be out of range here.  The
            # launch was flagged in the 'dim_size' style, so whoever meets the flag raises the
            # reference's ValueError (nn/aggr/base.py:131-141): this call when the flag has already
            # arrived (no wait), else the next scatter / aggregation call, the backward, or
            # `check_index_errors()`.  Only PYGAMD_CHECK_INDEX=sync blocks the host here — with
            # the default 'async' a per-aggregation wait would serialise host and device in every
            # unfused conv layer (MessagePassing.aggregate always supplies `dim_size`).
            if given and x.is_cuda and not torch.cuda.is_current_stream_capturing():
                _native.poll_index_errors(wait=_native.INDEX_CHECK == 'sync',
                                          device=x.device)
            return out
        except (IndexError, RuntimeError) as e:  # same recovery as nn/aggr/base.py:131-141
            if index is not None and index.numel() > 0:
                hi = _native.index_minmax(index)[1]
                if dim_size <= hi:
                    raise ValueError(f"Encountered invalid 'dim_size' (got "
                                     f"'{dim_size}' but expected "
                                     f">= '{hi + 1}')")
            raise e
        finally:
            _native.set_error_style(prev)

    def __repr__(self) -> str:
        return f'{self.__class__.__name__}()'

    def reduce(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
               dim_size: Optional[int] = None, dim: int = -2, reduce: str = 'sum') -> Tensor:
        if ptr is not None and index is None:
            d = dim + x.dim() if dim < 0 else dim
            if d != 0:
                return segment(x.movedim(d, 0).contiguous(), ptr, reduce).movedim(0, d)
            return segment(x, ptr, reduce=reduce)
        if index is None:
            raise RuntimeError("Aggregation requires 'index' to be specified")
        return scatter(x, index, dim, dim_size, reduce)

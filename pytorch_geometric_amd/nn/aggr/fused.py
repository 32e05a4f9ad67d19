import math
from typing import Dict, List, Optional, Union

import torch
from torch import Tensor

from ...utils import scatter
from .base import Aggregation
from .basic import (MaxAggregation, MeanAggregation, MinAggregation, MulAggregation,
                    SumAggregation, aggregation_resolver)


class VarAggregation(Aggregation):
    r"""``mean(x^2) - mean(x)^2`` per group (torch_geometric/nn/aggr/basic.py:82-112)."""

    def __init__(self, semi_grad: bool = False):
        super().__init__()
        self.semi_grad = semi_grad

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        mean = self.reduce(x, index, ptr, dim_size, dim, reduce='mean')
        if self.semi_grad:
            with torch.no_grad():
                mean2 = self.reduce(x * x, index, ptr, dim_size, dim, 'mean')
        else:
            mean2 = self.reduce(x * x, index, ptr, dim_size, dim, 'mean')
        return mean2 - mean * mean


class StdAggregation(Aggregation):
    r"""``sqrt(var)`` with the reference's clamp at 1e-5 (basic.py:115-139)."""

    def __init__(self, semi_grad: bool = False):
        super().__init__()
        self.var_aggr = VarAggregation(semi_grad)

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        var = self.var_aggr(x, index, ptr, dim_size, dim)
        out = var.clamp(min=1e-5).sqrt()
        return out.masked_fill(out <= math.sqrt(1e-5), 0.0)


_REDUCE = {'SumAggregation': 'sum', 'MeanAggregation': 'sum', 'MinAggregation': 'min',
           'MaxAggregation': 'max', 'MulAggregation': 'mul', 'VarAggregation': 'pow_sum',
           'StdAggregation': 'pow_sum'}
_FUSABLE = (SumAggregation, MeanAggregation, MinAggregation, MaxAggregation, MulAggregation,
            VarAggregation, StdAggregation)
_DEGREE_BASED = (MeanAggregation, VarAggregation, StdAggregation)


def _resolve(aggr) -> Aggregation:
    if isinstance(aggr, str) and aggr.lower() in ('var', 'std'):
        return VarAggregation() if aggr.lower() == 'var' else StdAggregation()
    return aggregation_resolver(aggr)


class FusedAggregation(Aggregation):
    r"""Several simple aggregations in one call, sharing the group count, the sum and the sum of
    squares between them (torch_geometric/nn/aggr/fused.py:191-336): ``mean`` reuses ``sum``,
    ``var`` reuses ``mean``/``sum``, ``std`` reuses ``var``.  Returns one tensor per aggregation,
    in the order given."""

    def __init__(self, aggrs: List[Union[Aggregation, str]]):
        super().__init__()
        if not isinstance(aggrs, (list, tuple)):
            raise ValueError(f"'aggrs' of '{self.__class__.__name__}' should "
                             f"be a list or tuple (got '{type(aggrs)}').")
        if len(aggrs) == 0:
            raise ValueError(f"'aggrs' of '{self.__class__.__name__}' should not be empty.")
        mods = [_resolve(a) for a in aggrs]
        for m in mods:
            if not isinstance(m, _FUSABLE):
                raise ValueError(f"Received aggregation '{m.__class__.__name__}' in "
                                 f"'{self.__class__.__name__}' which is not fusable")
        self.aggr_names = [m.__class__.__name__ for m in mods]
        self.aggr_index: Dict[str, int] = {n: i for i, n in enumerate(self.aggr_names)}
        self.semi_grad = any(getattr(m, 'semi_grad', False) or
                             getattr(getattr(m, 'var_aggr', None), 'semi_grad', False)
                             for m in mods)
        self.need_degree = any(isinstance(m, _DEGREE_BASED) for m in mods)

    def _one_pass(self, x: Tensor, index: Tensor, dim_size: Optional[int]) -> Dict[str, Tensor]:
        """sum / sum of squares / min / max / count of every group from ONE read of the rows: the
        index is sorted once (stable radix sort; skipped when it already is), then a single
        multi-reduce kernel walks the groups (csrc/spmm.hip ``spmm_multi_rows``)."""
        from ... import _native
        from ..._functions import MultiReduceFunction
        need = {'SumAggregation': ('sum', ), 'MeanAggregation': ('sum', ),
                'MinAggregation': ('min', ), 'MaxAggregation': ('max', ),
                'VarAggregation': ('sum', 'pow_sum'), 'StdAggregation': ('sum', 'pow_sum')}
        want = tuple(k for k in ('sum', 'pow_sum', 'min', 'max')
                     if any(k in need.get(n, ()) for n in self.aggr_names))
        if not want:
            return {}
        n = index.numel()
        lo, hi = _native.index_minmax(index) if n > 0 else (0, -1)
        if dim_size is None:
            dim_size = hi + 1
        if lo < 0 or hi >= dim_size:
            raise IndexError(f'index {hi if hi >= dim_size else lo} is out of bounds for '
                             f'dimension 0 with size {dim_size}')
        index = index.contiguous()
        if n > 1 and bool((index[1:] >= index[:-1]).all()):
            key, perm = index, None
        else:
            key, perm = _native.index_sort(index, max_value=max(dim_size - 1, 0))
        ptr = _native.index2ptr(key, dim_size)
        outs = MultiReduceFunction.apply(x.contiguous(), index, ptr, perm, want, self.semi_grad)
        cache = dict(zip(want, outs))
        cache['count'] = (ptr[1:] - ptr[:-1]).clamp(min=1).to(torch.float32).view(-1, 1)
        return cache

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> List[Tensor]:
        if index is None:
            raise NotImplementedError("Aggregation requires 'index' to be specified")
        if x.dim() != 2:
            raise ValueError(f"Aggregation requires two-dimensional inputs (got '{x.dim()}')")
        if dim not in (-2, 0):
            raise ValueError(f"Aggregation needs to perform aggregation in first dimension "
                             f"(got '{dim}')")
        cache: Dict[str, Tensor] = {}
        if x.is_cuda and x.dtype == torch.float32:
            cache.update(self._one_pass(x, index, dim_size))

        def get(reduce: str) -> Tensor:
            if reduce not in cache:
                if reduce == 'pow_sum':
                    src = x.detach() * x.detach() if self.semi_grad else x * x
                    cache[reduce] = scatter(src, index, 0, dim_size, 'sum')
                elif reduce == 'count':
                    ones = x.new_ones(x.size(0), 1)
                    cache[reduce] = scatter(ones, index, 0, dim_size, 'sum').clamp_(min=1)
                else:
                    cache[reduce] = scatter(x, index, 0, dim_size, reduce)
            return cache[reduce]

        def mean() -> Tensor:
            return get('sum') / get('count')

        def var() -> Tensor:
            m = mean()
            return get('pow_sum') / get('count') - m * m

        outs: List[Tensor] = []
        for name in self.aggr_names:
            if name == 'MeanAggregation':
                outs.append(mean())
            elif name == 'VarAggregation':
                outs.append(var())
            elif name == 'StdAggregation':
                s = var().clamp(min=1e-5).sqrt()
                outs.append(s.masked_fill(s <= math.sqrt(1e-5), 0.0))
            else:
                outs.append(get(_REDUCE[name]))
        return outs


class MultiAggregation(Aggregation):
    r"""Runs several aggregations and combines them (torch_geometric/nn/aggr/multi.py); modes
    ``cat`` (default), ``sum``, ``mean``, ``max``, ``min``.  Fusable members go through
    :class:`FusedAggregation`."""

    def __init__(self, aggrs: List[Union[Aggregation, str]], mode: Optional[str] = 'cat'):
        super().__init__()
        if not isinstance(aggrs, (list, tuple)) or len(aggrs) == 0:
            raise ValueError("'aggrs' should be a non-empty list or tuple")
        if mode not in ('cat', 'sum', 'mean', 'max', 'min'):
            raise ValueError(f"'Combine mode '{mode}' is not supported")
        self.aggrs = torch.nn.ModuleList([_resolve(a) for a in aggrs])
        self.mode = mode
        self.is_fused = [isinstance(a, _FUSABLE) for a in self.aggrs]
        fused = [a for a, f in zip(self.aggrs, self.is_fused) if f]
        self.fused_aggr = FusedAggregation(fused) if len(fused) > 1 else None

    def get_out_channels(self, in_channels: int) -> int:
        return in_channels * len(self.aggrs) if self.mode == 'cat' else in_channels

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        if index is None or x.dim() != 2 or self.fused_aggr is None:
            outs = [a(x, index, ptr, dim_size, dim) for a in self.aggrs]
        else:
            fused_outs = iter(self.fused_aggr(x, index, ptr, dim_size, dim))
            outs = [next(fused_outs) if f else a(x, index, ptr, dim_size, dim)
                    for a, f in zip(self.aggrs, self.is_fused)]
        if len(outs) == 1:
            return outs[0]
        if self.mode == 'cat':
            return torch.cat(outs, dim=-1)
        stacked = torch.stack(outs, dim=0)
        if self.mode == 'sum':
            return stacked.sum(0)
        if self.mode == 'mean':
            return stacked.mean(0)
        return stacked.max(0).values if self.mode == 'max' else stacked.min(0).values

"""Sum / Mean / Max / Min / Mul aggregations (torch_geometric/nn/aggr/basic.py:19-79)."""
from typing import Optional

from torch import Tensor

from .base import Aggregation


class SumAggregation(Aggregation):
    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        return self.reduce(x, index, ptr, dim_size, dim, reduce='sum')


class MeanAggregation(Aggregation):
    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        return self.reduce(x, index, ptr, dim_size, dim, reduce='mean')


class MaxAggregation(Aggregation):
    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        return self.reduce(x, index, ptr, dim_size, dim, reduce='max')


class MinAggregation(Aggregation):
    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        return self.reduce(x, index, ptr, dim_size, dim, reduce='min')


class MulAggregation(Aggregation):
    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        # like the reference, only `index` is supported for "mul"
        if index is None:
            raise NotImplementedError("Aggregation requires 'index' to be specified")
        return self.reduce(x, index, None, dim_size, dim, reduce='mul')


_BY_NAME = {
    'sum': SumAggregation, 'add': SumAggregation, 'mean': MeanAggregation,
    'max': MaxAggregation, 'min': MinAggregation, 'mul': MulAggregation,
}


def aggregation_resolver(aggr, **kwargs) -> Aggregation:
    """String / module -> :class:`Aggregation` (torch_geometric/nn/resolver.py role)."""
    if isinstance(aggr, Aggregation):
        return aggr
    if isinstance(aggr, str) and aggr.lower() in ('softmax', 'powermean'):
        from . import deeper
        by_name = {"softmax": deeper.SoftmaxAggregation, "powermean": deeper.PowerMeanAggregation}
        cls = by_name[aggr.lower()]
        return cls(**kwargs)
    if isinstance(aggr, str) and aggr.lower() in _BY_NAME:
        return _BY_NAME[aggr.lower()](**kwargs)
    raise ValueError(f"Could not resolve aggregation '{aggr}' "
                     f"(supported: {sorted(_BY_NAME)})")

"""The two learnable-temperature aggregations of DeeperGCN — contracts of
``SoftmaxAggregation`` / ``PowerMeanAggregation`` (torch_geometric/nn/aggr/basic.py:142-296).
Both are thin compositions of the path's primitives: the segment softmax kernel + a sum reduce,
and a mean reduce between two ``clamp().pow()`` maps."""
from typing import Optional, Union

import torch
from torch import Tensor
from torch.nn import Parameter

from ...utils import softmax
from .base import Aggregation


def _first_dim_2d(x: Tensor, dim: int) -> None:
    # per-channel parameters broadcast over [rows, channels] only (aggr/base.py:162-169)
    if x.dim() != 2:
        raise ValueError(f"Aggregation requires two-dimensional inputs (got '{x.dim()}')")
    if dim not in (-2, 0):
        raise ValueError(f"Aggregation needs to perform aggregation in first dimension "
                         f"(got '{dim}')")


class _Tempered(Aggregation):
    """Shared handling of the scalar-or-learnable exponent / temperature."""

    def _setup(self, name: str, value: float, learn: bool, channels: int) -> None:
        if channels != 1 and not learn:
            raise ValueError(f"Cannot set 'channels' greater than '1' in case "
                             f"'{type(self).__name__}' is not trainable")
        self._name, self._initial = name, value
        self.learn, self.channels = learn, channels
        setattr(self, name, Parameter(torch.empty(channels)) if learn else value)
        self.reset_parameters()

    def reset_parameters(self):
        value = getattr(self, self._name)
        if isinstance(value, Tensor):
            value.data.fill_(self._initial)

    def _value(self, x: Tensor, dim: int) -> Union[float, Tensor]:
        value = getattr(self, self._name)
        if self.channels != 1:
            _first_dim_2d(x, dim)
            value = value.view(-1, self.channels)
        return value

    def __repr__(self) -> str:
        return f'{type(self).__name__}(learn={self.learn})'


class SoftmaxAggregation(_Tempered):
    r"""``sum_i softmax_i(t * x_i) * x_i`` per group; ``t`` fixed or learned (optionally per
    channel); ``semi_grad`` treats the softmax weights as constants in the backward."""

    def __init__(self, t: float = 1.0, learn: bool = False, semi_grad: bool = False,
                 channels: int = 1):
        super().__init__()
        if learn and semi_grad:
            raise ValueError(f"Cannot enable 'semi_grad' in '{type(self).__name__}' in "
                             f"case the temperature term 't' is learnable")
        self.semi_grad = semi_grad
        self._setup('t', t, learn, channels)

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        t = self._value(x, dim)
        logits = x if (isinstance(t, (int, float)) and t == 1) else x * t
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.semi_grad):
            weights = softmax(logits, index, ptr, dim_size, dim)
        return self.reduce(x * weights, index, ptr, dim_size, dim, reduce='sum')


class PowerMeanAggregation(_Tempered):
    r"""``(mean_i x_i^p)^(1/p)`` per group with both the inputs and the mean clamped to
    ``[clamp_min, clamp_max]``; ``p`` fixed or learned (optionally per channel)."""

    def __init__(self, p: float = 1.0, learn: bool = False, channels: int = 1,
                 clamp_min: Optional[float] = 1e-4, clamp_max: Optional[float] = 100.):
        super().__init__()
        self.min_value, self.max_value = clamp_min, clamp_max
        self._setup('p', p, learn, channels)

    def forward(self, x: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
                dim_size: Optional[int] = None, dim: int = -2) -> Tensor:
        p = self._value(x, dim)
        plain = isinstance(p, (int, float)) and p == 1
        if not plain:
            x = x.clamp(min=self.min_value, max=self.max_value).pow(p)
        out = self.reduce(x, index, ptr, dim_size, dim, reduce='mean')
        if not plain:
            out = out.clamp(min=self.min_value, max=self.max_value).pow(1. / p)
        return out

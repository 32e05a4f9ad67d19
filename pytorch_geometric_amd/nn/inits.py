"""Parameter initialisers: the distributions of torch_geometric/nn/inits.py:8-60 (uniform,
kaiming-uniform, glorot, constants), applied to plain tensors / parameters; ``None`` is ignored so
optional parameters can be passed unconditionally."""
import math
from typing import Optional

from torch import Tensor


def _symmetric_uniform_(tensor: Optional[Tensor], bound: float) -> None:
    if tensor is not None:
        tensor.data.uniform_(-bound, bound)


def uniform(size: int, value: Optional[Tensor]) -> None:
    """U(-1/sqrt(size), 1/sqrt(size))."""
    _symmetric_uniform_(value, 1.0 / math.sqrt(size))


def kaiming_uniform(value: Optional[Tensor], fan: int, a: float) -> None:
    """He-uniform with negative slope ``a`` and the given fan."""
    _symmetric_uniform_(value, math.sqrt(6.0 / ((1.0 + a * a) * fan)))


def glorot(value: Optional[Tensor]) -> None:
    """Xavier-uniform over the last two dimensions."""
    if value is not None:
        fan_sum = value.size(-2) + value.size(-1)
        _symmetric_uniform_(value, math.sqrt(6.0 / fan_sum))


def constant(value: Optional[Tensor], fill_value: float) -> None:
    if value is not None:
        value.data.fill_(fill_value)


def zeros(value: Optional[Tensor]) -> None:
    constant(value, 0.0)


def ones(value: Optional[Tensor]) -> None:
    constant(value, 1.0)

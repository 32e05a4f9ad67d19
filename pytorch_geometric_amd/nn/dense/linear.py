import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.parameter import Parameter

from .. import inits


class Linear(torch.nn.Module):
    r"""``x @ W.T + b`` with the reference's initialisers
    (torch_geometric/nn/dense/linear.py:59-127: ``weight_initializer`` in ``glorot | uniform |
    kaiming_uniform | None``, ``bias_initializer`` in ``zeros | None``; ``None`` matches
    :class:`torch.nn.Linear`).  The GEMM is a plain library call (rocBLAS / hipBLASLt through
    ``F.linear``) — MFMA-bound, not part of the HBM-bound aggregation path."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True,
                 weight_initializer: Optional[str] = None,
                 bias_initializer: Optional[str] = None):
        super().__init__()
        if in_channels <= 0:
            raise ValueError("lazy initialisation (in_channels=-1) is not supported")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.weight_initializer = weight_initializer
        self.bias_initializer = bias_initializer
        self.weight = Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == 'glorot':
            inits.glorot(self.weight)
        elif self.weight_initializer == 'uniform':
            bound = 1.0 / math.sqrt(self.weight.size(-1))
            torch.nn.init.uniform_(self.weight.data, -bound, bound)
        elif self.weight_initializer == 'kaiming_uniform' or self.weight_initializer is None:
            inits.kaiming_uniform(self.weight, fan=self.in_channels, a=math.sqrt(5))
        else:
            raise RuntimeError(f"Linear layer weight initializer "
                               f"'{self.weight_initializer}' is not supported")
        if self.bias is not None:
            if self.bias_initializer == 'zeros':
                inits.zeros(self.bias)
            elif self.bias_initializer is None:
                inits.uniform(self.in_channels, self.bias)
            else:
                raise RuntimeError(f"Linear layer bias initializer "
                                   f"'{self.bias_initializer}' is not supported")

    def forward(self, x: Tensor) -> Tensor:
        return F.linear(x, self.weight, self.bias)

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, '
                f'bias={self.bias is not None})')

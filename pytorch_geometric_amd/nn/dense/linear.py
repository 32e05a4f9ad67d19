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
    :class:`torch.nn.Linear`).  float32 HIP inputs with at least ``OWN_GEMM_MIN_ROWS`` rows run on
    this repo's fp32-MFMA GEMM (csrc/gemm.hip through
    :class:`~pytorch_geometric_amd._functions.LinearFunction`: exact fp32, an ``fmaf`` chain per
    output); anything else (small inputs, other dtypes, CPU tensors during module construction /
    ``state_dict`` round trips in the tests) is ``F.linear``."""

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

    # From this many rows up the own kernels run (csrc/gemm.hip): 128-row tiles when the rows alone
    # fill the chip, otherwise 64 x 64 tiles and a deterministic split over the reduction (GCN on
    # the Cora shape, 2,708 rows x 1,433 features: 264 workgroups instead of the 22 of round 3's
    # row-tiled launch).  Below it: F.linear.
    from ..._functions import OWN_GEMM_MIN_ROWS

    def forward(self, x: Tensor) -> Tensor:
        from ..._functions import linear
        return linear(x, self.weight, self.bias)

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, '
                f'bias={self.bias is not None})')


def hetero_linear_forward(mod, x: Tensor, type_vec: Tensor) -> Tensor:
    """``HeteroLinear.forward`` (nn/dense/linear.py:287-329): sort the rows by type (stable HIP
    radix sort) unless ``is_sorted``, ONE grouped fp32-MFMA GEMM (``segment_matmul``), the per-type
    bias, the original order restored.  ``mod``: this package's layer or the reference's (same
    parameter names and shapes, linear.py:213-246) through ``backend.install()``."""
    from ... import _native
    from ..._functions import GatherFunction
    from ...utils import segment_matmul
    perm = None
    if type_vec.numel() > 0:
        # (the grouped GEMM reads the segment pointer on the host anyway: one more tiny sync)
        lo, hi = _native.index_minmax(type_vec)
        if lo < 0 or hi >= mod.num_types:
            raise IndexError(f"'type_vec' must lie in [0, {mod.num_types}) "
                             f"(got values in [{lo}, {hi}])")
    if not mod.is_sorted:
        type_vec, perm = _native.index_sort(type_vec, max_value=mod.num_types)
        x = GatherFunction.apply(x, perm, False)
    type_ptr = _native.index2ptr(type_vec, mod.num_types)
    out = segment_matmul(x, type_ptr, mod.weight)
    if mod.bias is not None:
        out = out + GatherFunction.apply(mod.bias, type_vec, False)
    if perm is not None:  # restore the original order
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        out = GatherFunction.apply(out, inv, False)
    return out


class HeteroLinear(torch.nn.Module):
    r"""One linear map per node/edge type: ``out[k] = x[k] @ W[type_vec[k]] + b[type_vec[k]]`` —
    constructor, parameters (``weight [T, in, out]``, ``bias [T, out]``) and forward of
    ``torch_geometric.nn.HeteroLinear`` (torch_geometric/nn/dense/linear.py:170-329).  Rows are
    stably sorted by type (HIP radix sort), transformed by ONE grouped fp32-MFMA GEMM
    (``segment_matmul``) and restored to their original order."""

    def __init__(self, in_channels: int, out_channels: int, num_types: int,
                 is_sorted: bool = False, bias: bool = True,
                 weight_initializer: Optional[str] = None,
                 bias_initializer: Optional[str] = None):
        super().__init__()
        if in_channels <= 0:
            raise ValueError("lazy initialisation (in_channels=-1) is not supported")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_types = num_types
        self.is_sorted = is_sorted
        self.weight_initializer = weight_initializer
        self.bias_initializer = bias_initializer
        self.weight = Parameter(torch.empty(num_types, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.empty(num_types, out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == 'glorot':
            inits.glorot(self.weight)
        elif self.weight_initializer == 'uniform':
            bound = 1.0 / math.sqrt(self.in_channels)
            torch.nn.init.uniform_(self.weight.data, -bound, bound)
        elif self.weight_initializer in ('kaiming_uniform', None):
            inits.kaiming_uniform(self.weight, fan=self.in_channels, a=math.sqrt(5))
        else:
            raise RuntimeError(f"Weight initializer '{self.weight_initializer}' not supported")
        if self.bias is not None:
            if self.bias_initializer == 'zeros':
                inits.zeros(self.bias)
            elif self.bias_initializer is None:
                inits.uniform(self.in_channels, self.bias)
            else:
                raise RuntimeError(f"Bias initializer '{self.bias_initializer}' not supported")

    def forward(self, x: Tensor, type_vec: Tensor) -> Tensor:
        return hetero_linear_forward(self, x, type_vec)

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, '
                f'num_types={self.num_types}, bias={self.bias is not None})')

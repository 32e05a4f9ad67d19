from typing import Optional, Tuple, Union

import torch.nn.functional as F
from torch import Tensor

from ..._functions import SpmmFunction
from ...edge_index import EdgeIndex
from ..dense.linear import Linear
from .message_passing import MessagePassing


class SAGEConv(MessagePassing):
    r"""GraphSAGE operator ``x_i' = W_1 x_i + W_2 * mean_{j in N(i)} x_j`` — same constructor,
    parameters (``lin_l``, ``lin_r``, optional ``lin``) and forward semantics as
    ``torch_geometric.nn.SAGEConv`` (torch_geometric/nn/conv/sage_conv.py:68-152).

    The neighbourhood reduction runs at the INPUT width as one CSR SpMM launch
    (``message_and_aggregate``); the two linear maps are library GEMMs.
    """

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 aggr: str = 'mean', normalize: bool = False, root_weight: bool = True,
                 project: bool = False, bias: bool = True, **kwargs):
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalize = normalize
        self.root_weight = root_weight
        self.project = project
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        super().__init__(aggr, **kwargs)
        if self.project:
            if in_channels[0] <= 0:
                raise ValueError(f"'{self.__class__.__name__}' does not support lazy "
                                 f"initialization with `project=True`")
            self.lin = Linear(in_channels[0], in_channels[0], bias=True)
        self.lin_l = Linear(in_channels[0], out_channels, bias=bias)
        if self.root_weight:
            self.lin_r = Linear(in_channels[1], out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        if self.project:
            self.lin.reset_parameters()
        self.lin_l.reset_parameters()
        if self.root_weight:
            self.lin_r.reset_parameters()

    def forward(self, x: Union[Tensor, Tuple[Tensor, Optional[Tensor]]], edge_index,
                size: Optional[Tuple[int, int]] = None) -> Tensor:
        if isinstance(x, Tensor):
            x = (x, x)
        if self.project and hasattr(self, 'lin'):
            x = (self.lin(x[0]).relu(), x[1])
        out = self.propagate(edge_index, x=x, size=size)
        out = self.lin_l(out)
        x_r = x[1]
        if self.root_weight and x_r is not None:
            out = out + self.lin_r(x_r)
        if self.normalize:
            out = F.normalize(out, p=2., dim=-1)
        return out

    def message(self, x_j: Tensor) -> Tensor:
        return x_j

    def message_and_aggregate(self, graph: EdgeIndex, x) -> Tensor:
        return SpmmFunction.apply(x[0], None, graph, 'sum' if self.aggr == 'add' else self.aggr,
                                  'coo')

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_channels}, '
                f'{self.out_channels}, aggr={self.aggr})')

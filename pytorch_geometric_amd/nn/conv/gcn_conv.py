from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.nn import Parameter

from ..._functions import SpmmFunction
from ...edge_index import EdgeIndex
from ...utils import add_remaining_self_loops, scatter
from ...utils.num_nodes import maybe_num_nodes
from ..dense.linear import Linear
from ..inits import zeros
from .message_passing import MessagePassing


def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor] = None,
             num_nodes: Optional[int] = None, improved: bool = False,
             add_self_loops: bool = True, flow: str = 'source_to_target',
             dtype: Optional[torch.dtype] = None) -> Tuple[Tensor, Tensor]:
    r"""Symmetric normalisation ``D^-1/2 (A + I) D^-1/2`` on a COO edge list — same steps as
    torch_geometric/nn/conv/gcn_conv.py:45-113 (tensor branch): add the remaining self-loops with
    weight ``fill_value``, degree by scatter-add over the destination, ``deg^-1/2`` with inf -> 0,
    per-edge weight ``dis[row] * w * dis[col]``."""
    fill_value = 2. if improved else 1.
    assert flow in ('source_to_target', 'target_to_source')
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value,
                                                           num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1), ), dtype=dtype or torch.float32,
                                 device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    idx = col if flow == 'source_to_target' else row
    deg = scatter(edge_weight, idx, dim=0, dim_size=num_nodes, reduce='sum')
    deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt = deg_inv_sqrt.masked_fill(deg_inv_sqrt == float('inf'), 0)
    edge_weight = deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]
    return edge_index, edge_weight


class GCNConv(MessagePassing):
    r"""Graph convolution ``X' = D^-1/2 (A + I) D^-1/2 X W + b`` — constructor, parameters
    (``lin`` with glorot init, ``bias`` zeros) and forward order (normalise -> transform ->
    propagate at the OUTPUT width -> bias) of ``torch_geometric.nn.GCNConv``
    (torch_geometric/nn/conv/gcn_conv.py:116-274)."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False,
                 cached: bool = False, add_self_loops: Optional[bool] = None,
                 normalize: bool = True, bias: bool = True, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        if add_self_loops is None:
            add_self_loops = normalize
        if add_self_loops and not normalize:
            raise ValueError(f"'{self.__class__.__name__}' does not support "
                             f"adding self-loops to the graph when no "
                             f"on-the-fly normalization is applied")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached
        self.add_self_loops = add_self_loops
        self.normalize = normalize
        self._cached_edge_index = None
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer='glorot')
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        self.lin.reset_parameters()
        zeros(self.bias)
        self._cached_edge_index = None

    def forward(self, x: Tensor, edge_index, edge_weight: Optional[Tensor] = None) -> Tensor:
        if isinstance(x, (tuple, list)):
            raise ValueError(f"'{self.__class__.__name__}' received a tuple "
                             f"of node features as input while this layer "
                             f"does not support bipartite message passing. "
                             f"Please try other layers such as 'SAGEConv' or "
                             f"'GraphConv' instead")
        if self.normalize and isinstance(edge_index, Tensor):
            cache = self._cached_edge_index
            if cache is None:
                edge_index, edge_weight = gcn_norm(edge_index, edge_weight,
                                                   x.size(self.node_dim), self.improved,
                                                   self.add_self_loops, self.flow, x.dtype)
                if self.cached:
                    self._cached_edge_index = (edge_index, edge_weight)
            else:
                edge_index, edge_weight = cache[0], cache[1]
        x = self.lin(x)
        out = self.propagate(edge_index, x=x, edge_weight=edge_weight)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def message_and_aggregate(self, graph: EdgeIndex, x: Tensor,
                              edge_weight: Optional[Tensor]) -> Tensor:
        return SpmmFunction.apply(x, edge_weight, graph, 'sum', 'coo')

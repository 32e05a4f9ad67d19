from typing import Optional, Tuple, Union

from torch import Tensor

from ..._functions import SpmmFunction, spmm_node
from ...edge_index import EdgeIndex
from ..dense.linear import Linear
from .message_passing import MessagePassing


class GraphConv(MessagePassing):
    r"""``x_i' = W_root x_i + W_rel * aggr_{j in N(i)} e_{ji} x_j`` (Morris et al., k-GNN) with the
    constructor arguments and parameter names (``lin_rel`` with bias, ``lin_root`` without) of
    ``torch_geometric.nn.GraphConv`` (torch_geometric/nn/conv/graph_conv.py:49-112) — the one
    reference layer that already fuses on a sorted ``EdgeIndex`` (``SUPPORTS_FUSED_EDGE_INDEX``).
    Here any ``edge_index`` takes the fused, optionally edge-weighted, CSR SpMM."""

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 aggr: str = 'add', bias: bool = True, **kwargs):
        super().__init__(aggr=aggr, **kwargs)
        src_dim, dst_dim = ((in_channels, in_channels) if isinstance(in_channels, int)
                            else tuple(in_channels))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_rel = Linear(src_dim, out_channels, bias=bias)
        self.lin_root = Linear(dst_dim, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def forward(self, x, edge_index, edge_weight: Optional[Tensor] = None,
                size: Optional[Tuple[int, int]] = None) -> Tensor:
        if isinstance(x, Tensor) and edge_weight is None:
            from ..models import _fused_sage
            if _fused_sage.layer_eligible(self, x, edge_index, size):
                # the SAGE layer under other names: aggregation + both maps + bias in one kernel
                return _fused_sage.run_layer(self, x, edge_index)
        pair = (x, x) if isinstance(x, Tensor) else x
        out = self.lin_rel(self.propagate(edge_index, x=pair, edge_weight=edge_weight,
                                          size=size))
        return out if pair[1] is None else out + self.lin_root(pair[1])

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def _can_fuse(self, kwargs) -> bool:
        if kwargs.get('edge_weight') is not None and self.aggr in ('min', 'max'):
            return False  # weighted extrema take the general gather / scatter route
        return super()._can_fuse(kwargs)

    def message_and_aggregate(self, graph: EdgeIndex, x, edge_weight) -> Tensor:
        reduce = {'add': 'sum'}.get(self.aggr, self.aggr)
        return spmm_node(x[0], edge_weight, graph, reduce, 'coo')

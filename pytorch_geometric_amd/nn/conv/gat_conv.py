import weakref
from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import Parameter

from ..._functions import GatEdgeSoftmaxFunction, SpmmFunction
from ...edge_index import EdgeIndex, as_edge_index
from ...utils import add_self_loops, remove_self_loops, softmax
from ..dense.linear import Linear
from ..inits import glorot, zeros
from .message_passing import MessagePassing


class GATConv(MessagePassing):
    r"""Graph attention operator — constructor, parameters (``lin`` / ``lin_src`` / ``lin_dst``,
    ``att_src``, ``att_dst``, optional ``lin_edge`` / ``att_edge`` / ``res``, ``bias``) and forward
    semantics of ``torch_geometric.nn.GATConv`` (torch_geometric/nn/conv/gat_conv.py:130-413):
    ``alpha_ij = softmax_j(leaky_relu(a_src . W x_j + a_dst . W x_i))``, messages
    ``alpha_ij * W x_j`` summed per destination, heads concatenated or averaged.

    Fused path (no ``edge_attr``): ONE kernel builds the logits from the two ``[N, H]`` node terms
    and normalises them per destination row (never materialising the gathered logits), ONE
    multi-head weighted SpMM aggregates ``[N, H*C]`` rows.
    """

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 heads: int = 1, concat: bool = True, negative_slope: float = 0.2,
                 dropout: float = 0.0, add_self_loops: bool = True,
                 edge_dim: Optional[int] = None, fill_value: Union[float, Tensor, str] = 'mean',
                 bias: bool = True, residual: bool = False, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(node_dim=0, **kwargs)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.heads = heads
        self.concat = concat
        self.negative_slope = negative_slope
        self.dropout = dropout
        self.add_self_loops = add_self_loops
        self.edge_dim = edge_dim
        self.fill_value = fill_value
        self.residual = residual
        self.lin = self.lin_src = self.lin_dst = None
        if isinstance(in_channels, int):
            self.lin = Linear(in_channels, heads * out_channels, bias=False,
                              weight_initializer='glorot')
        else:
            self.lin_src = Linear(in_channels[0], heads * out_channels, False,
                                  weight_initializer='glorot')
            self.lin_dst = Linear(in_channels[1], heads * out_channels, False,
                                  weight_initializer='glorot')
        self.att_src = Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = Parameter(torch.empty(1, heads, out_channels))
        if edge_dim is not None:
            self.lin_edge = Linear(edge_dim, heads * out_channels, bias=False,
                                   weight_initializer='glorot')
            self.att_edge = Parameter(torch.empty(1, heads, out_channels))
        else:
            self.lin_edge = None
            self.register_parameter('att_edge', None)
        total_out_channels = out_channels * (heads if concat else 1)
        if residual:
            self.res = Linear(in_channels if isinstance(in_channels, int) else in_channels[1],
                              total_out_channels, bias=False, weight_initializer='glorot')
        else:
            self.register_parameter('res', None)
        if bias:
            self.bias = Parameter(torch.empty(total_out_channels))
        else:
            self.register_parameter('bias', None)
        self._loop_cache = None
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        for lin in (self.lin, self.lin_src, self.lin_dst, self.lin_edge, self.res):
            if lin is not None:
                lin.reset_parameters()
        glorot(self.att_src)
        glorot(self.att_dst)
        glorot(self.att_edge)
        zeros(self.bias)

    def _with_self_loops(self, edge_index: Tensor, edge_attr: Optional[Tensor], num_nodes: int):
        """remove_self_loops + add_self_loops (gat_conv.py:334-346).  Without edge features the
        result is cached per input tensor so the sorted handle of the augmented graph is reused
        across forward calls."""
        if edge_attr is None:
            hit = self._loop_cache
            if hit is not None:
                ref, version, n, out = hit
                if ref() is edge_index and version == edge_index._version and n == num_nodes:
                    return out, None
        ei, ea = remove_self_loops(edge_index, edge_attr)
        ei, ea = add_self_loops(ei, ea, fill_value=self.fill_value, num_nodes=num_nodes)
        if edge_attr is None:
            self._loop_cache = (weakref.ref(edge_index), edge_index._version, num_nodes, ei)
        return ei, ea

    def forward(self, x: Union[Tensor, Tuple[Tensor, Optional[Tensor]]], edge_index,
                edge_attr: Optional[Tensor] = None, size: Optional[Tuple[int, int]] = None,
                return_attention_weights: Optional[bool] = None):
        H, C = self.heads, self.out_channels
        res: Optional[Tensor] = None
        if isinstance(x, Tensor):
            assert x.dim() == 2, "Static graphs not supported in 'GATConv'"
            if self.res is not None:
                res = self.res(x)
            if self.lin is not None:
                x_src = x_dst = self.lin(x).view(-1, H, C)
            else:
                assert self.lin_src is not None and self.lin_dst is not None
                x_src = self.lin_src(x).view(-1, H, C)
                x_dst = self.lin_dst(x).view(-1, H, C)
        else:
            x_src, x_dst = x
            assert x_src.dim() == 2, "Static graphs not supported in 'GATConv'"
            if x_dst is not None and self.res is not None:
                res = self.res(x_dst)
            if self.lin is not None:
                x_src = self.lin(x_src).view(-1, H, C)
                if x_dst is not None:
                    x_dst = self.lin(x_dst).view(-1, H, C)
            else:
                assert self.lin_src is not None and self.lin_dst is not None
                x_src = self.lin_src(x_src).view(-1, H, C)
                if x_dst is not None:
                    x_dst = self.lin_dst(x_dst).view(-1, H, C)
        x = (x_src, x_dst)
        alpha_src = (x_src * self.att_src).sum(dim=-1)
        alpha_dst = None if x_dst is None else (x_dst * self.att_dst).sum(-1)
        alpha = (alpha_src, alpha_dst)

        if self.add_self_loops and isinstance(edge_index, Tensor):
            num_nodes = x_src.size(0)
            if x_dst is not None:
                num_nodes = min(num_nodes, x_dst.size(0))
            num_nodes = min(size) if size is not None else num_nodes
            edge_index, edge_attr = self._with_self_loops(edge_index, edge_attr, num_nodes)

        fused = (self.fuse and edge_attr is None and alpha_dst is not None
                 and self.flow == 'source_to_target')
        if fused:
            n_src = x_src.size(0)
            n_dst = x_dst.size(0) if size is None else size[1]
            graph = as_edge_index(edge_index, n_src, n_dst)
            alpha_slot = GatEdgeSoftmaxFunction.apply(alpha_src, alpha_dst[:n_dst].contiguous(),
                                                      graph, self.negative_slope)
            att = F.dropout(alpha_slot, p=self.dropout, training=self.training)
            out = SpmmFunction.apply(x_src.reshape(n_src, H * C), att, graph, 'sum', 'slot')
            out = out.view(-1, H, C)
            alpha_out = None
            if return_attention_weights is not None:
                alpha_out = torch.empty_like(alpha_slot)
                alpha_out[graph.by_dst().perm.long()] = alpha_slot
        else:
            alpha_out = self.edge_updater(edge_index, alpha=alpha, edge_attr=edge_attr,
                                          size=size)
            fuse, self.fuse = self.fuse, False
            try:
                out = self.propagate(edge_index, x=x, alpha=alpha_out, size=size)
            finally:
                self.fuse = fuse

        if self.concat:
            out = out.view(-1, self.heads * self.out_channels)
        else:
            out = out.mean(dim=1)
        if res is not None:
            out = out + res
        if self.bias is not None:
            out = out + self.bias
        if return_attention_weights is not None:
            ei = edge_index.edge_index if isinstance(edge_index, EdgeIndex) else edge_index
            return out, (ei, alpha_out)
        return out

    def edge_update(self, alpha_j: Tensor, alpha_i: Optional[Tensor],
                    edge_attr: Optional[Tensor], index: Tensor, ptr: Optional[Tensor],
                    dim_size: Optional[int]) -> Tensor:
        alpha = alpha_j if alpha_i is None else alpha_j + alpha_i
        if index.numel() == 0:
            return alpha
        if edge_attr is not None and self.lin_edge is not None:
            if edge_attr.dim() == 1:
                edge_attr = edge_attr.view(-1, 1)
            edge_attr = self.lin_edge(edge_attr)
            edge_attr = edge_attr.view(-1, self.heads, self.out_channels)
            alpha = alpha + (edge_attr * self.att_edge).sum(dim=-1)
        alpha = F.leaky_relu(alpha, self.negative_slope)
        alpha = softmax(alpha, index, ptr, dim_size)
        return F.dropout(alpha, p=self.dropout, training=self.training)

    def message(self, x_j: Tensor, alpha: Tensor) -> Tensor:
        return alpha.unsqueeze(-1) * x_j

    def message_and_aggregate(self, graph: EdgeIndex, x, alpha) -> Tensor:
        raise NotImplementedError  # fusion is driven from forward() (needs slot-ordered alpha)

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_channels}, '
                f'{self.out_channels}, heads={self.heads})')

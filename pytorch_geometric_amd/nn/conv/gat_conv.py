import weakref
from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import Parameter

from ..._functions import (GatAttendFunction, GatEdgeSoftmaxFunction, HeadDotFunction,
                           SpmmFunction, bias_act)
from ...edge_index import EdgeIndex, as_edge_index
from ...utils import add_self_loops, remove_self_loops, softmax
from ..dense.linear import Linear
from ..inits import glorot, zeros
from ._act_request import requested_activation
from .message_passing import MessagePassing


def _glorot_linear(n_in: int, n_out: int) -> Linear:
    return Linear(n_in, n_out, bias=False, weight_initializer='glorot')


class GATConv(MessagePassing):
    r"""Graph attention operator with the constructor arguments, parameter names (``lin`` or
    ``lin_src`` / ``lin_dst``, ``att_src``, ``att_dst``, optional ``lin_edge`` / ``att_edge`` /
    ``res``, ``bias``) and forward semantics of ``torch_geometric.nn.GATConv``
    (torch_geometric/nn/conv/gat_conv.py:130-413):

    .. math:: \alpha_{ij} = \mathrm{softmax}_j\,\mathrm{LeakyReLU}(a_s^\top W x_j + a_d^\top W x_i),
              \qquad x_i' = \big\Vert_h \sum_j \alpha^h_{ij} W^h x_j \;(\text{or the head mean}).

    Fused path (no edge features): ONE kernel builds the logits from the two ``[N, H]`` node terms
    and normalises them per destination row (the gathered logits are never materialised), ONE
    multi-head weighted SpMM aggregates the ``[N, H*C]`` rows.  With ``edge_dim`` or
    ``fuse = False`` the layer takes the general gather -> ``edge_update`` -> scatter route.
    """

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 heads: int = 1, concat: bool = True, negative_slope: float = 0.2,
                 dropout: float = 0.0, add_self_loops: bool = True,
                 edge_dim: Optional[int] = None, fill_value: Union[float, Tensor, str] = 'mean',
                 bias: bool = True, residual: bool = False, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(node_dim=0, **kwargs)
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops, self.edge_dim = add_self_loops, edge_dim
        self.fill_value, self.residual = fill_value, residual
        width = heads * out_channels
        shared = isinstance(in_channels, int)
        # one shared projection for homogeneous inputs, two for (source, destination) pairs
        self.lin = _glorot_linear(in_channels, width) if shared else None
        self.lin_src = None if shared else _glorot_linear(in_channels[0], width)
        self.lin_dst = None if shared else _glorot_linear(in_channels[1], width)
        self.att_src = Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = Parameter(torch.empty(1, heads, out_channels))
        self.lin_edge = None if edge_dim is None else _glorot_linear(edge_dim, width)
        self.register_parameter(
            'att_edge', None if edge_dim is None else Parameter(torch.empty(1, heads,
                                                                             out_channels)))
        out_width = width if concat else out_channels
        dst_in = in_channels if shared else in_channels[1]
        self.res = _glorot_linear(dst_in, out_width) if residual else None
        self.register_parameter('bias', Parameter(torch.empty(out_width)) if bias else None)
        self._loop_cache = None
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        for lin in (self.lin, self.lin_src, self.lin_dst, self.lin_edge, self.res):
            if lin is not None:
                lin.reset_parameters()
        for att in (self.att_src, self.att_dst, self.att_edge):
            glorot(att)
        zeros(self.bias)

    # -- pieces of forward -------------------------------------------------------------------------
    def _project(self, x):
        """(x_src [N_s, H, C], x_dst [N_d, H, C] | None, residual | None)."""
        H, C = self.heads, self.out_channels
        pair = not isinstance(x, Tensor)
        raw_src, raw_dst = x if pair else (x, x)
        assert raw_src.dim() == 2, "Static graphs not supported in 'GATConv'"
        res = self.res(raw_dst) if (self.res is not None and raw_dst is not None) else None
        f_src = self.lin if self.lin is not None else self.lin_src
        f_dst = self.lin if self.lin is not None else self.lin_dst
        x_src = f_src(raw_src).view(-1, H, C)
        if raw_dst is None:
            x_dst = None
        elif not pair and self.lin is not None:
            x_dst = x_src  # same tensor, same projection
        else:
            x_dst = f_dst(raw_dst).view(-1, H, C)
        return x_src, x_dst, res

    def _with_self_loops(self, edge_index: Tensor, edge_attr: Optional[Tensor], num_nodes: int):
        """Drop existing self-loops, then add one per node (gat_conv.py:334-346).  Without edge
        features the augmented edge list is cached per input tensor, so the sorted handle built
        for it is reused by later forward calls."""
        if edge_attr is None and self._loop_cache is not None:
            ref, version, n, cached = self._loop_cache
            if ref() is edge_index and version == edge_index._version and n == num_nodes:
                return cached, None
        ei, ea = remove_self_loops(edge_index, edge_attr)
        ei, ea = add_self_loops(ei, ea, fill_value=self.fill_value, num_nodes=num_nodes)
        if edge_attr is None:
            self._loop_cache = (weakref.ref(edge_index), edge_index._version, num_nodes, ei)
        return ei, ea

    def forward(self, x: Union[Tensor, Tuple[Tensor, Optional[Tensor]]], edge_index,
                edge_attr: Optional[Tensor] = None, size: Optional[Tuple[int, int]] = None,
                return_attention_weights: Optional[bool] = None):
        H, C = self.heads, self.out_channels
        x_src, x_dst, res = self._project(x)
        # The kernels compute in float32 (CHANGELOG.md §4).  Half / bf16 projected features — a half
        # model, or an autocast region whose `Linear` ran in bf16 — are widened HERE and take the
        # same native path; outside autocast the result is handed back in their dtype.  (They used
        # to reach float32-only kernels as they were and raise: ADVICE r4.)
        low = None
        if x_src.is_cuda and x_src.dtype in (torch.float16, torch.bfloat16):
            low = x_src.dtype
            same = x_dst is x_src
            x_src = x_src.float()
            x_dst = x_src if same else (None if x_dst is None else x_dst.float())
            res = None if res is None else res.float()
        att_src, att_dst = self.att_src, self.att_dst
        if att_src.dtype != x_src.dtype:
            att_src, att_dst = att_src.to(x_src.dtype), att_dst.to(x_src.dtype)
        native = x_src.is_cuda and x_src.dtype == torch.float32 and self.fuse
        # A single-head layer on a handle marked `atomic_backward` (a sampled batch, used once)
        # keeps the HeadDot + SpmmFunction route, whose backward runs edge-parallel atomics instead
        # of building the by-source sort the fused node's backward needs (the atomic backward
        # exists for one weight per edge: heads == 1).
        one_shot = (isinstance(edge_index, EdgeIndex) and edge_index.atomic_backward
                    and self.heads == 1)
        # one autograd node for node terms + edge softmax + aggregation (GatAttendFunction) when
        # nothing between them is observable: no edge features, no dropout on the coefficients,
        # nobody asking for them
        attend = (x_dst is x_src and native and edge_attr is None and not one_shot
                  and self.flow == 'source_to_target' and return_attention_weights is None
                  and not (self.training and self.dropout > 0))
        if attend:
            a_src = a_dst = None  # computed inside the fused node
        elif x_dst is x_src and native:
            a_src, a_dst = HeadDotFunction.apply(x_src, att_src, att_dst)
        else:
            a_src = (x_src * att_src).sum(dim=-1)
            a_dst = None if x_dst is None else (x_dst * att_dst).sum(dim=-1)

        if self.add_self_loops and isinstance(edge_index, Tensor) \
                and not isinstance(edge_index, EdgeIndex):
            n = x_src.size(0) if x_dst is None else min(x_src.size(0), x_dst.size(0))
            n = min(size) if size is not None else n
            edge_index, edge_attr = self._with_self_loops(edge_index, edge_attr, n)
        elif self.add_self_loops and isinstance(edge_index, EdgeIndex):
            # handles get the same remove + add self-loops treatment as tensors (the reference
            # takes one branch for both, gat_conv.py:334-347); the result is cached on the handle
            handle = edge_index
            n = min(handle.sparse_size) if size is None else min(size)

            def build():
                ei, ea = remove_self_loops(handle.edge_index, edge_attr)
                ei, ea = add_self_loops(ei, ea, fill_value=self.fill_value, num_nodes=n)
                out = EdgeIndex(ei, handle.sparse_size, validate=False)
                out.atomic_backward = handle.atomic_backward
                return out, ea

            if edge_attr is None:
                edge_index, edge_attr = handle.derived(('self_loops', n), build)
            else:
                edge_index, edge_attr = build()

        use_fused = (native and edge_attr is None and a_dst is not None
                     and self.flow == 'source_to_target')
        if attend and edge_attr is None:
            n_src = x_src.size(0)
            n_dst = n_src if size is None else size[1]
            graph = as_edge_index(edge_index, n_src, n_dst)
            out = GatAttendFunction.apply(x_src, att_src, att_dst, graph,
                                          self.negative_slope, n_dst)
            alpha = None
        elif attend:  # (the self-loop rewrite produced edge attributes: cannot happen without
            raise AssertionError('edge attributes appeared on the fused GAT path')  # input ones)
        elif use_fused:
            n_src = x_src.size(0)
            n_dst = x_dst.size(0) if size is None else size[1]
            graph = as_edge_index(edge_index, n_src, n_dst)
            alpha_slot = GatEdgeSoftmaxFunction.apply(a_src, a_dst[:n_dst].contiguous(), graph,
                                                      self.negative_slope)
            weights = F.dropout(alpha_slot, p=self.dropout, training=self.training)
            out = SpmmFunction.apply(x_src.reshape(n_src, H * C), weights, graph, 'sum', 'slot')
            out = out.view(-1, H, C)
            alpha = None
            if return_attention_weights is not None:
                # the POST-dropout coefficients (what edge_update returns in the reference,
                # gat_conv.py:404-406), back in the caller's edge order
                alpha = torch.empty_like(weights)
                alpha[graph.by_dst().perm.long()] = weights
        else:
            alpha = self.edge_updater(edge_index, alpha=(a_src, a_dst), edge_attr=edge_attr,
                                      size=size)
            keep, self.fuse = self.fuse, False
            try:
                out = self.propagate(edge_index, x=(x_src, x_dst), alpha=alpha, size=size)
            finally:
                self.fuse = keep

        out = out.reshape(-1, H * C) if self.concat else out.mean(dim=1)
        if res is not None:
            out = out + res
        # a ReLU stack's request (BasicGNN, _act_request): bias + the model's activation in one pass
        fa = requested_activation(self)
        if fa is not None or (self.bias is not None and out.is_cuda):
            out = bias_act(out, self.bias, fa == 'relu')
        elif self.bias is not None:
            out = out + self.bias
        if low is not None and not torch.is_autocast_enabled():
            out = out.to(low)
        if return_attention_weights is None:
            return out
        coo = edge_index.edge_index if isinstance(edge_index, EdgeIndex) else edge_index
        return out, (coo, alpha)

    def edge_update(self, alpha_j: Tensor, alpha_i: Optional[Tensor],
                    edge_attr: Optional[Tensor], index: Tensor, ptr: Optional[Tensor],
                    dim_size: Optional[int]) -> Tensor:
        logits = alpha_j if alpha_i is None else alpha_j + alpha_i
        if index.numel() == 0:
            return logits
        if edge_attr is not None and self.lin_edge is not None:
            e = edge_attr.view(-1, 1) if edge_attr.dim() == 1 else edge_attr
            e = self.lin_edge(e).view(-1, self.heads, self.out_channels)
            logits = logits + (e * self.att_edge).sum(dim=-1)
        att = softmax(F.leaky_relu(logits, self.negative_slope), index, ptr, dim_size)
        return F.dropout(att, p=self.dropout, training=self.training)

    def message(self, x_j: Tensor, alpha: Tensor) -> Tensor:
        return alpha.unsqueeze(-1) * x_j

    def message_and_aggregate(self, graph: EdgeIndex, x, alpha) -> Tensor:
        raise NotImplementedError  # fusion is driven from forward() (needs slot-ordered alpha)

    def __repr__(self) -> str:
        return (f'{type(self).__name__}({self.in_channels}, {self.out_channels}, '
                f'heads={self.heads})')

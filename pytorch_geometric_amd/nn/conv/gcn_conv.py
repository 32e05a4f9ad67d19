from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.nn import Parameter

from ..._functions import SpmmFunction, spmm_node, bias_act
from ...edge_index import EdgeIndex
from ...utils import add_remaining_self_loops, scatter
from ...utils.num_nodes import maybe_num_nodes
from ..dense.linear import Linear
from ..inits import zeros
from ._act_request import requested_activation
from .message_passing import MessagePassing


def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor] = None,
             num_nodes: Optional[int] = None, improved: bool = False,
             add_self_loops: bool = True, flow: str = 'source_to_target',
             dtype: Optional[torch.dtype] = None) -> Tuple[Tensor, Tensor]:
    r"""``D^-1/2 (A + c I) D^-1/2`` on a COO edge list, ``c = 2`` if ``improved`` else ``1`` — the
    tensor branch of torch_geometric/nn/conv/gcn_conv.py:45-113: remaining self-loops are added
    with weight ``c``, the degree is the scatter-add of the weights over the receiving endpoint,
    isolated nodes get ``deg^-1/2 = 0``, and every edge is scaled by both endpoint factors."""
    if flow not in ('source_to_target', 'target_to_source'):
        raise ValueError(f"invalid flow '{flow}'")
    n = maybe_num_nodes(edge_index, num_nodes)
    if add_self_loops:
        loop_weight = 2.0 if improved else 1.0
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, loop_weight, n)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype or torch.float32,
                                 device=edge_index.device)
    receiver = edge_index[1] if flow == 'source_to_target' else edge_index[0]
    inv_sqrt = scatter(edge_weight, receiver, dim=0, dim_size=n, reduce='sum').pow(-0.5)
    inv_sqrt = torch.where(torch.isinf(inv_sqrt), torch.zeros_like(inv_sqrt), inv_sqrt)
    return edge_index, inv_sqrt[edge_index[0]] * edge_weight * inv_sqrt[edge_index[1]]


_FLOW_HOOKS = ('_propagate_forward_pre_hooks', '_propagate_forward_hooks',
               '_message_forward_pre_hooks', '_message_forward_hooks',
               '_aggregate_forward_pre_hooks', '_aggregate_forward_hooks',
               '_message_and_aggregate_forward_pre_hooks',
               '_message_and_aggregate_forward_hooks', '_edge_update_forward_pre_hooks',
               '_edge_update_forward_hooks')


def linear_message_flow(conv, base) -> bool:
    """``A (X W) = (A X) W`` holds for this layer object: a LINEAR aggregation (``add`` / ``sum`` /
    ``mean`` given as a string — ``GCNConv(16, 64, aggr='max')`` is a valid layer and is not),
    the stock ``message`` / ``aggregate`` / ``message_and_aggregate`` / ``update`` of ``base`` (a
    subclass that overrides one of them inherits ``forward`` too), and nobody observing the flow
    (hooks, explain mode, decomposed layers).  ``base``: the ``GCNConv`` class ``conv`` derives
    from — this package's or, through ``backend.install()``, the reference's."""
    if not (isinstance(conv.aggr, str) and conv.aggr in ('add', 'sum', 'mean')):
        return False
    kind = type(conv)
    for name in ('message', 'aggregate', 'message_and_aggregate', 'update'):
        if getattr(kind, name, None) is not getattr(base, name, None):
            return False
    if getattr(conv, 'explain', False) or getattr(conv, 'decomposed_layers', 1) != 1:
        return False
    return not any(getattr(conv, name, None) for name in _FLOW_HOOKS)


class GCNConv(MessagePassing):
    r"""Graph convolution ``X' = D^-1/2 (A + I) D^-1/2 X W + b`` with the constructor arguments,
    parameters (``lin.weight`` glorot, ``bias`` zeros) and the normalise -> transform -> propagate
    (at the OUTPUT width) -> bias order of ``torch_geometric.nn.GCNConv``
    (torch_geometric/nn/conv/gcn_conv.py:116-274).  ``cached=True`` keeps the normalised edge list,
    and with it the sorted graph handle, across calls."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False,
                 cached: bool = False, add_self_loops: Optional[bool] = None,
                 normalize: bool = True, bias: bool = True, **kwargs):
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        add_self_loops = normalize if add_self_loops is None else add_self_loops
        if add_self_loops and not normalize:
            raise ValueError(f"'{type(self).__name__}' does not support adding self-loops to "
                             f"the graph when no on-the-fly normalization is applied")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached = improved, cached
        self.add_self_loops, self.normalize = add_self_loops, normalize
        self._cached_edge_index = None
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer='glorot')
        self.register_parameter('bias', Parameter(torch.empty(out_channels)) if bias else None)
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        self.lin.reset_parameters()
        zeros(self.bias)
        self._cached_edge_index = None

    def _normalized(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor]):
        if self._cached_edge_index is not None:
            return self._cached_edge_index
        out = gcn_norm(edge_index, edge_weight, x.size(self.node_dim), self.improved,
                       self.add_self_loops, self.flow, x.dtype)
        if self.cached:
            self._cached_edge_index = out
        return out

    def _normalized_handle(self, x: Tensor, handle: EdgeIndex, edge_weight: Optional[Tensor]):
        n = x.size(self.node_dim)
        if handle.sparse_size != (n, n):
            raise ValueError(f"'{type(self).__name__}' needs a square graph over the {n} rows "
                             f"of 'x' (got sparse_size={handle.sparse_size})")

        def build():
            ei, w = gcn_norm(handle.edge_index, edge_weight, n, self.improved,
                             self.add_self_loops, self.flow, x.dtype)
            return EdgeIndex(ei, (n, n), validate=False), w

        if edge_weight is not None:  # values (and maybe a graph of gradients) differ per call
            return build()
        return handle.derived(('gcn_norm', self.improved, self.add_self_loops, self.flow), build)

    def forward(self, x: Tensor, edge_index, edge_weight: Optional[Tensor] = None) -> Tensor:
        if isinstance(x, (tuple, list)):
            raise ValueError(f"'{type(self).__name__}' received a tuple of node features as "
                             f"input while this layer does not support bipartite message "
                             f"passing. Please try other layers such as 'SAGEConv' or "
                             f"'GraphConv' instead")
        if self.normalize and isinstance(edge_index, EdgeIndex):
            # the reference normalises its EdgeIndex inputs too (gcn_conv.py:241-258): never
            # aggregate an un-normalised handle silently.  Checked first: a handle IS a Tensor
            edge_index, edge_weight = self._normalized_handle(x, edge_index, edge_weight)
        elif self.normalize and isinstance(edge_index, Tensor):
            edge_index, edge_weight = self._normalized(x, edge_index, edge_weight)
        if self._aggregate_first(x):
            # A (X W) = (A X) W: aggregate at the narrower width.  The reference always transforms
            # first (gcn_conv.py:260-264); a 100 -> 256 layer then gathers 2.5 x the bytes, and
            # its backward needs the transposed aggregation even when `x` takes no gradient.
            out = self.lin(self.propagate(edge_index, x=x, edge_weight=edge_weight))
        else:
            out = self.propagate(edge_index, x=self.lin(x), edge_weight=edge_weight)
        # a ReLU stack's request (BasicGNN, _act_request): bias + the model's activation in one pass
        fa = requested_activation(self)
        if fa is not None or (self.bias is not None and out.is_cuda):
            return bias_act(out, self.bias, fa == 'relu')
        return out if self.bias is None else out + self.bias

    def _aggregate_first(self, x: Tensor) -> bool:
        """Aggregation before the linear map: only where it is the same computation seen from
        outside (float32 device features on the fused route, a linear aggregation with the
        stock message, nobody hooked into the message flow: :func:`linear_message_flow`) and the
        input is the narrower side.  ``aggregate_first = False`` on the layer keeps
        the reference's order."""
        if not getattr(self, 'aggregate_first', True) or not self.fuse:
            return False
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
            return False
        if x.size(-1) >= self.out_channels:
            return False
        return linear_message_flow(self, GCNConv)

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def message_and_aggregate(self, graph: EdgeIndex, x: Tensor,
                              edge_weight: Optional[Tensor]) -> Tensor:
        # (the layer's OWN aggregation: `GCNConv(..., aggr='mean' | 'max')` are valid layers,
        # gcn_conv.py:181 only sets a default)
        return spmm_node(x, edge_weight, graph, {'add': 'sum'}.get(self.aggr, self.aggr), 'coo')

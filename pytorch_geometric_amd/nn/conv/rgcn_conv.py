import weakref
from typing import Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn import Parameter

from ... import _native
from ..._functions import GatherFunction, SpmmFunction, bias_act, linear
from ...edge_index import EdgeIndex
from ...utils._segment_matmul import segment_matmul_sum
from ..inits import glorot, zeros
from ._act_request import requested_activation
from .message_passing import MessagePassing


class RelationalHandle:
    r"""Per-(graph, edge_type) cache for :class:`RGCNConv`: the edges stably sorted by
    ``(relation, destination)`` so that every non-empty ``(r, i)`` pair is one contiguous segment.

    * ``pair_graph``  — handle over ``[E]`` edges -> ``[S]`` pair segments (rows = pairs, columns =
      source nodes): one SpMM gives the per-relation neighbourhood mean ``agg[s]``;
    * ``rel_ptr``     — ``[R+1]`` ranges of pairs per relation (pairs are sorted by relation): the
      segments of ``segment_matmul``;
    * ``out_graph``   — handle pairs -> destination nodes: one SpMM (sum) scatters the transformed
      pair rows back onto the nodes.

    Everything is built with the same bit-exact sort / pointer kernels as :class:`EdgeIndex`."""

    def __init__(self, edge_index: Tensor, edge_type: Tensor, num_src: int, num_dst: int,
                 num_relations: int):
        dev = edge_index.device
        src, dst = edge_index[0], edge_index[1]
        if edge_type.numel() > 0:
            # the radix sort below only looks at the bits `max_value` needs: a relation id outside
            # [0, num_relations) would be mis-sorted silently (the reference fails with an index
            # error in its per-relation loop).  Once per handle (handles are cached).
            lo, hi = _native.index_minmax(edge_type)
            if lo < 0 or hi >= num_relations:
                raise IndexError(f"'edge_type' must lie in [0, {num_relations}) "
                                 f"(got values in [{lo}, {hi}])")
            lo, hi = _native.index_minmax(dst)
            if lo < 0 or hi >= num_dst:
                raise IndexError(f"Found indices in 'edge_index' outside the valid range "
                                 f"[0, {num_dst - 1}] (got interval [{lo}, {hi}])")
        key = edge_type.to(torch.int64) * num_dst + dst.to(torch.int64)
        skey, perm = _native.index_sort(key, max_value=num_relations * num_dst)
        # pair boundaries: positions where the sorted key changes
        E = key.numel()
        if E > 0:
            new = torch.ones(E, dtype=torch.bool, device=dev)
            new[1:] = skey[1:] != skey[:-1]
            starts = new.nonzero().view(-1)
            pair_key = skey[starts]
            seg_ptr = torch.cat([starts, torch.tensor([E], device=dev)])
        else:
            pair_key = skey
            seg_ptr = torch.zeros(1, dtype=torch.int64, device=dev)
        S = pair_key.numel()
        self.num_pairs = S
        pair_rel = torch.div(pair_key, num_dst, rounding_mode='floor')
        pair_dst = pair_key - pair_rel * num_dst
        src_sorted = _native.permute_index(src.to(torch.int64), perm)
        # edges -> pairs (rows = pairs).  The COO form is only needed for the transposed handle.
        pair_of_edge = _native.ptr2index(seg_ptr, E)
        self.pair_graph = EdgeIndex(torch.stack([src_sorted, pair_of_edge]), (num_src, S),
                                    sort_order='col', validate=False)
        self._perm, self._pair_of_edge, self._edge_graph = perm, pair_of_edge, None
        self.rel_ptr = tuple(_native.index2ptr(pair_rel, num_relations).tolist())
        # the relation histogram's maximum (what the reference's `use_segment_matmul_heuristic`
        # looks at, rgcn_conv.py:246-257): pairs per relation are on the host already, the edges
        # per pair are `seg_ptr`'s differences — one more small read, once per handle
        if E > 0:
            edge_ptr = seg_ptr[torch.tensor(self.rel_ptr, device=dev)]
            self.max_edges_per_relation = int((edge_ptr[1:] - edge_ptr[:-1]).max())
        else:
            self.max_edges_per_relation = 0
        # pairs -> destination nodes
        self.out_graph = EdgeIndex(torch.stack([torch.arange(S, device=dev), pair_dst]),
                                   (S, num_dst), sort_order='row', validate=False)


    @property
    def edge_graph(self) -> EdgeIndex:
        """Handle over per-EDGE message rows -> pair segments (rows = pairs, columns = the
        original edge ids): aggregates messages that were built edge by edge (node-index
        inputs).  Built on first use."""
        if self._edge_graph is None:
            E = self._perm.numel()
            self._edge_graph = EdgeIndex(torch.stack([self._perm, self._pair_of_edge]),
                                         (E, self.num_pairs), sort_order='col', validate=False)
        return self._edge_graph


def _relational_handle(conv, edge_index: Tensor, edge_type: Tensor, n_src: int,
                       n_dst: int) -> RelationalHandle:
    """The layer's handle for this ``(edge_index, edge_type)`` pair, cached on the layer by tensor
    identity + in-place version (one sort per graph, not per call).  ``conv`` is duck-typed: the
    reference's ``RGCNConv`` objects are served through ``backend.install()``."""
    hit = conv.__dict__.get('_handle_cache')
    if hit is not None:
        r_ei, r_et, v_ei, v_et, size, handle = hit
        if (r_ei() is edge_index and r_et() is edge_type and v_ei == edge_index._version
                and v_et == edge_type._version and size == (n_src, n_dst)):
            return handle
    handle = RelationalHandle(edge_index, edge_type, n_src, n_dst, conv.num_relations)
    conv.__dict__['_handle_cache'] = (weakref.ref(edge_index), weakref.ref(edge_type),
                                      edge_index._version, edge_type._version, (n_src, n_dst),
                                      handle)
    return handle


def _index_inputs(conv, x_l, x_r, edge_index, edge_type, weight, by_node_id=False):
    """Node-index ("featureless") inputs, rgcn_conv.py:262-268: the message of edge (j, i, r)
    is the embedding row ``weight[r, x_l[j]]``.  One differentiable row gather builds all
    messages, one SpMM reduces them per (relation, destination) pair and one sums the pairs
    of every destination — instead of one masked propagate per relation."""
    ei = edge_index.edge_index if isinstance(edge_index, EdgeIndex) else edge_index
    n_src, n_dst = x_l.size(0), x_r.size(0)
    look = ei[0] if by_node_id else x_l[ei[0]]
    rows = edge_type.to(torch.int64) * weight.size(1) + look.to(torch.int64)
    msg = GatherFunction.apply(weight.reshape(-1, conv.out_channels), rows, True)
    h = _relational_handle(conv, ei, edge_type, n_src, n_dst)
    reduce = 'sum' if conv.aggr == 'add' else conv.aggr
    per_pair = SpmmFunction.apply(msg, None, h.edge_graph, reduce, 'coo')
    return SpmmFunction.apply(per_pair, None, h.out_graph, 'sum', 'coo')


def _composed_weight(conv) -> Tensor:
    weight = conv.weight
    if conv.num_bases is not None:  # basis decomposition (rgcn_conv.py:203-205)
        weight = (conv.comp @ weight.view(conv.num_bases, -1)).view(
            conv.num_relations, conv.in_channels_l, conv.out_channels)
    return weight


def _root_bias(conv, out: Tensor, x_r: Tensor) -> Tensor:
    root = conv.root
    if root is not None:
        if not torch.is_floating_point(x_r):
            out = out + root[x_r]
        else:
            out = out + linear(x_r, root.t())
    fa = requested_activation(conv)  # (BasicGNN-style ReLU stacks: bias + ReLU in one pass)
    if fa is not None or (conv.bias is not None and out.is_cuda):
        return bias_act(out, conv.bias, fa == 'relu')
    if conv.bias is not None:
        out = out + conv.bias
    return out


def rgcn_forward(conv, x, edge_index, edge_type: Optional[Tensor]) -> Tensor:
    """``RGCNConv.forward`` (rgcn_conv.py:164-282, the per-relation loop's result) on the sorted,
    segmented schedule of the class docstring below.  ``conv``: this package's layer or the
    reference's (same parameter names and shapes, rgcn_conv.py:92-162)."""
    x_l = x[0] if isinstance(x, tuple) else x
    if x_l is None:
        x_l = torch.arange(conv.in_channels_l, device=conv.weight.device)
    x_r = x[1] if isinstance(x, tuple) else x_l
    assert edge_type is not None
    if not torch.is_floating_point(x_r) and conv.num_blocks is not None:
        raise ValueError('Block-diagonal decomposition not supported '
                         'for non-continuous input features.')
    weight = _composed_weight(conv)
    if not torch.is_floating_point(x_l):
        # node-index inputs: the per-relation embedding lookup weight[r, x_j]
        out = _index_inputs(conv, x_l, x_r, edge_index, edge_type, weight)
    else:
        n_src, n_dst = x_l.size(0), x_r.size(0)
        ei = edge_index.edge_index if isinstance(edge_index, EdgeIndex) else edge_index
        h = _relational_handle(conv, ei, edge_type, n_src, n_dst)
        reduce = 'sum' if conv.aggr == 'add' else conv.aggr
        # (1) per-(relation, destination) neighbourhood reduce at the input width
        agg = SpmmFunction.apply(x_l, None, h.pair_graph, reduce, 'coo')  # [S, F_in]
        # (2) relation-segmented transform (block b of a pair row times weight[r, b]: column
        # blocks in place, one launch) and (3) the sum of the pair rows of every destination,
        # as one autograd node: the backward of (3) is a row gather that the gradient launches
        # of (2) do themselves
        out = segment_matmul_sum(agg, h.rel_ptr, weight, h.out_graph)
    return _root_bias(conv, out, x_r)


def fast_rgcn_forward(conv, x, edge_index, edge_type: Optional[Tensor]) -> Tensor:
    """``FastRGCNConv.forward`` (rgcn_conv.py:301-374): the same operator restricted to ``aggr`` in
    ``add`` / ``sum`` / ``mean``; index inputs are looked up by SOURCE NODE id
    (rgcn_conv.py:357-359)."""
    assert conv.aggr in ['add', 'sum', 'mean']
    x_l = x[0] if isinstance(x, tuple) else x
    if x_l is not None and torch.is_floating_point(x_l):
        return rgcn_forward(conv, x, edge_index, edge_type)
    assert edge_type is not None
    if conv.num_blocks is not None:
        raise ValueError('Block-diagonal decomposition not supported '
                         'for non-continuous input features.')
    if x_l is None:
        x_l = torch.arange(conv.in_channels_l, device=conv.weight.device)
    x_r = x[1] if isinstance(x, tuple) else x_l
    out = _index_inputs(conv, x_l, x_r, edge_index, edge_type, _composed_weight(conv),
                        by_node_id=True)
    return _root_bias(conv, out, x_r)


class RGCNConv(MessagePassing):
    r"""Relational graph convolution
    ``x_i' = root x_i + sum_r mean_{j in N_r(i)} W_r x_j + b`` — constructor, parameters
    (``weight``, optional ``comp``, ``root``, ``bias``; glorot / zeros init) and semantics of the
    default branch of ``torch_geometric.nn.RGCNConv`` (torch_geometric/nn/conv/rgcn_conv.py:92-290),
    including ``num_bases`` (weights composed from bases) and ``num_blocks`` (block-diagonal).

    Instead of the reference's Python loop over relations (mask -> nonzero -> gather -> scatter ->
    matmul, 474 times for FB15k-237) the layer sorts the edges by ``(relation, destination)`` once
    (cached), aggregates every ``(r, i)`` neighbourhood with ONE SpMM, transforms the pair rows
    with a relation-segmented matmul and sums them per destination with a second SpMM.
    """

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 num_relations: int, num_bases: Optional[int] = None,
                 num_blocks: Optional[int] = None, aggr: str = 'mean', root_weight: bool = True,
                 is_sorted: bool = False, bias: bool = True, **kwargs):
        kwargs.setdefault('aggr', aggr)
        super().__init__(node_dim=0, **kwargs)
        if num_bases is not None and num_blocks is not None:
            raise ValueError('Can not apply both basis-decomposition and '
                             'block-diagonal-decomposition at the same time.')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_relations = num_relations
        self.num_bases = num_bases
        self.num_blocks = num_blocks
        self.is_sorted = is_sorted
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels_l = in_channels[0]
        if num_bases is not None:
            self.weight = Parameter(torch.empty(num_bases, in_channels[0], out_channels))
            self.comp = Parameter(torch.empty(num_relations, num_bases))
        elif num_blocks is not None:
            assert in_channels[0] % num_blocks == 0 and out_channels % num_blocks == 0
            self.weight = Parameter(torch.empty(num_relations, num_blocks,
                                                in_channels[0] // num_blocks,
                                                out_channels // num_blocks))
            self.register_parameter('comp', None)
        else:
            self.weight = Parameter(torch.empty(num_relations, in_channels[0], out_channels))
            self.register_parameter('comp', None)
        if root_weight:
            self.root = Parameter(torch.empty(in_channels[1], out_channels))
        else:
            self.register_parameter('root', None)
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        glorot(self.weight)
        glorot(self.comp)
        glorot(self.root)
        zeros(self.bias)

    def forward(self, x: Union[Optional[Tensor], Tuple[Optional[Tensor], Tensor]], edge_index,
                edge_type: Optional[Tensor] = None) -> Tensor:
        return rgcn_forward(self, x, edge_index, edge_type)

    def message(self, x_j: Tensor) -> Tensor:
        return x_j

    def message_and_aggregate(self, graph: EdgeIndex, x: Tensor) -> Tensor:
        return SpmmFunction.apply(x, None, graph, 'sum' if self.aggr == 'add' else self.aggr,
                                  'coo')

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_channels}, '
                f'{self.out_channels}, num_relations={self.num_relations})')


class FastRGCNConv(RGCNConv):
    r"""``torch_geometric.nn.FastRGCNConv`` (torch_geometric/nn/conv/rgcn_conv.py:301-374): the
    same operator as :class:`RGCNConv` restricted to ``aggr`` in ``add`` / ``sum`` / ``mean``.  The
    reference trades memory for speed there (one ``bmm`` over per-edge weight copies); here both
    classes already run the sorted, segmented path, so this class only adds the reference's
    argument checks and its node-index convention: index inputs are looked up by SOURCE NODE id
    (``rgcn_conv.py:357-359``), i.e. they are meaningful with ``x=None`` and
    ``in_channels == num_nodes``."""

    def forward(self, x, edge_index, edge_type: Optional[Tensor] = None) -> Tensor:
        return fast_rgcn_forward(self, x, edge_index, edge_type)

"""CPU oracle for the message-passing hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement of the reference's (PyG 2.9.0) CPU scatter path, function by function, in terms of
the same ATen CPU ops the reference bottoms out in (``index_select``, ``scatter_add_``,
``scatter_reduce_``, ``_segment_reduce``, stable ``sort``, ``_convert_indices_from_coo_to_csr``) —
SURVEY.md §8(c): "the arithmetic really lives in PyTorch ATen CPU kernels".  Every function cites
the reference file:line it follows (paths relative to the reference root).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
(``pytorch_geometric_amd``) never imports it.

Parity status: PINNED — ``tests/golden/make_golden.py`` runs the real reference
(``/root/reference``, importable in the build container) on seeded inputs and commits the
outputs; ``tests/test_oracle_golden.py`` checks every function here against those vectors and
against the reference's own known-answer tests.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


# ---- utils/_scatter.py -----------------------------------------------------------------------------
def broadcast(src: Tensor, ref: Tensor, dim: int) -> Tensor:
    """utils/_scatter.py:141-144."""
    dim = ref.dim() + dim if dim < 0 else dim
    shape = [1] * ref.dim()
    shape[dim] = -1
    return src.view(shape).expand_as(ref)


def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None,
            reduce: str = 'sum') -> Tensor:
    """utils/_scatter.py:14-138, CPU branches (no torch_scatter)."""
    if index.dim() != 1:
        raise ValueError(f"The `index` argument must be one-dimensional "
                         f"(got {index.dim()} dimensions)")
    dim = src.dim() + dim if dim < 0 else dim
    if dim < 0 or dim >= src.dim():
        raise ValueError(f"The `dim` argument must lay between 0 and "
                         f"{src.dim() - 1} (got {dim})")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    size = list(src.size())
    size[dim] = dim_size
    if reduce == 'any':                                            # :63-65
        return src.new_zeros(size).scatter_(dim, broadcast(index, src, dim), src)
    if reduce in ('sum', 'add'):                                   # :68-70
        return src.new_zeros(size).scatter_add_(dim, broadcast(index, src, dim), src)
    if reduce == 'mean':                                           # :72-80
        count = src.new_zeros(dim_size)
        count.scatter_add_(0, index, src.new_ones(src.size(dim)))
        count = count.clamp(min=1)
        out = src.new_zeros(size).scatter_add_(dim, broadcast(index, src, dim), src)
        return out / broadcast(count, out, dim)
    if reduce in ('min', 'max', 'amin', 'amax'):                   # :84-100
        return src.new_zeros(size).scatter_reduce_(dim, broadcast(index, src, dim), src,
                                                   reduce=f'a{reduce[-3:]}', include_self=False)
    if reduce == 'mul':                                            # :119-133
        return src.new_ones(size).scatter_reduce_(dim, broadcast(index, src, dim), src,
                                                  reduce='prod', include_self=True)
    raise ValueError(f"Encountered invalid `reduce` argument '{reduce}'")


def scatter_argmax(src: Tensor, index: Tensor, dim: int = 0,
                   dim_size: Optional[int] = None) -> Tensor:
    """utils/_scatter.py:147-184 (1-D only)."""
    assert src.dim() == 1 and index.dim() == 1
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    res = src.new_empty(dim_size)
    res.scatter_reduce_(0, index, src.detach(), reduce='amax', include_self=False)
    out = index.new_full((dim_size, ), fill_value=dim_size - 1)
    nonzero = (src == res[index]).nonzero().view(-1)
    out[index[nonzero]] = nonzero
    return out


# ---- index.py ------------------------------------------------------------------------------------
def ptr2index(ptr: Tensor, output_size: Optional[int] = None) -> Tensor:
    """index.py:27-29."""
    index = torch.arange(ptr.numel() - 1, dtype=ptr.dtype)
    return index.repeat_interleave(ptr.diff(), output_size=output_size)


def index2ptr(index: Tensor, size: Optional[int] = None) -> Tensor:
    """index.py:32-37."""
    if size is None:
        size = int(index.max()) + 1 if index.numel() > 0 else 0
    return torch._convert_indices_from_coo_to_csr(index, size,
                                                  out_int32=index.dtype != torch.int64)


def index_sort(inputs: Tensor, max_value: Optional[int] = None,
               stable: bool = True) -> Tuple[Tensor, Tensor]:
    """utils/_index_sort.py:10-32 without pyg-lib: ``inputs.sort(stable=...)``."""
    return inputs.sort(stable=stable)


def csr_from_coo(key: Tensor, other: Tensor, n_rows: int):
    """edge_index.py:589-623 (``get_indptr`` + ``_sort_by_transpose``): stable sort by ``key``;
    returns (ptr, other[perm], perm)."""
    sorted_key, perm = index_sort(key, max_value=n_rows, stable=True)
    return index2ptr(sorted_key, n_rows), other[perm], perm


# ---- utils/_segment.py ---------------------------------------------------------------------------
def segment(src: Tensor, ptr: Tensor, reduce: str = 'sum') -> Tensor:
    """utils/_segment.py:37-50 (``_torch_segment``)."""
    if reduce in ('min', 'max'):
        reduce = f'a{reduce}'
    initial = 0 if reduce == 'mean' else None
    out = torch._segment_reduce(src, reduce, offsets=ptr, initial=initial)
    if reduce in ('amin', 'amax'):
        out = torch.where(out.isinf(), 0, out)
    return out


# ---- utils/_softmax.py ---------------------------------------------------------------------------
def segment_logsumexp(src: Tensor, ptr: Tensor, dim: int) -> Tensor:
    """utils/_segment.py:53-80: scatter-max over ptr2index, exp of the shifted values, segment sum,
    log with -inf mapped to 0, plus the maximum."""
    src = src.transpose(0, dim)
    index = ptr2index(ptr, src.size(0))
    max_src = scatter(src, index, 0, ptr.numel() - 1, 'max')
    out = segment((src - max_src[index]).exp(), ptr, 'sum')
    out = out.log().nan_to_num(neginf=0.0) + max_src
    return out.transpose(0, dim)


def softmax(src: Tensor, index: Optional[Tensor] = None, ptr: Optional[Tensor] = None,
            num_nodes: Optional[int] = None, dim: int = 0) -> Tensor:
    """utils/_softmax.py:60-92 (ptr branch for 1-D ptr, else index branch)."""
    if ptr is not None and ptr.dim() == 1:
        dim = dim + src.dim() if dim < 0 else dim
        count = ptr[1:] - ptr[:-1]
        p = ptr.view([1] * dim + [-1])
        n = src.size(dim)
        src_max = segment(src.detach(), p, reduce='max')
        src_max = src_max.repeat_interleave(count, dim=dim, output_size=n)
        out = (src - src_max).exp()
        out_sum = segment(out, p, reduce='sum') + 1e-16
        out_sum = out_sum.repeat_interleave(count, dim=dim, output_size=n)
    elif index is not None:
        N = num_nodes if num_nodes is not None else (
            int(index.max()) + 1 if index.numel() > 0 else 0)
        src_max = scatter(src.detach(), index, dim, dim_size=N, reduce='max')
        out = (src - src_max.index_select(dim, index)).exp()
        out_sum = scatter(out, index, dim, dim_size=N, reduce='sum') + 1e-16
        out_sum = out_sum.index_select(dim, index)
    else:
        raise NotImplementedError("'softmax' requires 'index' to be specified")
    return out / out_sum


# ---- nn/aggr/basic.py:142-296 (DeeperGCN aggregations) ---------------------------------------------
def _group_reduce(x, index, ptr, dim_size, reduce):
    return segment(x, ptr, reduce) if ptr is not None else scatter(x, index, 0, dim_size, reduce)


def softmax_aggregation(x, index=None, ptr=None, dim_size=None, t=1.0, semi_grad=False):
    """nn/aggr/basic.py:196-215: alpha = softmax(x * t) per group, out = sum(x * alpha)."""
    logits = x * t
    if semi_grad:
        with torch.no_grad():
            alpha = softmax(logits, index, ptr, dim_size)
    else:
        alpha = softmax(logits, index, ptr, dim_size)
    return _group_reduce(x * alpha, index, ptr, dim_size, 'sum')


def powermean_aggregation(x, index=None, ptr=None, dim_size=None, p=1.0, lo=1e-4, hi=100.):
    """nn/aggr/basic.py:275-292: mean of clamp(x)^p, then clamp()^(1/p); p == 1 is a plain mean."""
    if isinstance(p, (int, float)) and p == 1:
        return _group_reduce(x, index, ptr, dim_size, 'mean')
    out = _group_reduce(x.clamp(min=lo, max=hi).pow(p), index, ptr, dim_size, 'mean')
    return out.clamp(min=lo, max=hi).pow(1. / p)


# ---- propagate = gather -> message -> scatter (nn/conv/message_passing.py:421-563) ------------------
def propagate(x_src: Tensor, edge_index: Tensor, num_dst: int, reduce: str,
              edge_weight: Optional[Tensor] = None) -> Tensor:
    """The unfused CPU scatter path: ``x_j = x.index_select(0, edge_index[0])``
    (message_passing.py:263-290), optional ``w_e * x_j`` message (gcn_conv.py:270-271; for a
    ``[E, H]`` weight and ``[N, H, C]`` features gat_conv.py:408-409), ``scatter`` onto
    ``edge_index[1]`` (nn/aggr/base.py:173-185)."""
    x_j = x_src.index_select(0, edge_index[0])
    if edge_weight is not None:
        if x_j.dim() == 3:
            x_j = edge_weight.unsqueeze(-1) * x_j
        else:
            x_j = edge_weight.view(-1, 1) * x_j
    reduce = 'sum' if reduce == 'add' else reduce
    return scatter(x_j, edge_index[1], 0, num_dst, reduce)


def spmm(edge_index: Tensor, x: Tensor, num_dst: int, reduce: str = 'sum',
         value: Optional[Tensor] = None) -> Tensor:
    """``_scatter_spmm`` (edge_index.py:1903-1922) in adj_t orientation."""
    return propagate(x, edge_index, num_dst, reduce, value)


# ---- utils/loop.py -------------------------------------------------------------------------------
def remove_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor] = None):
    """utils/loop.py:71."""
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], None if edge_attr is None else edge_attr[mask]


def _loop_attr(edge_index, edge_attr, num_nodes, fill_value):
    """utils/loop.py:742-769 (``compute_loop_attr``, non-sparse)."""
    size = (num_nodes, ) + tuple(edge_attr.size()[1:])
    if fill_value is None:
        return edge_attr.new_ones(size)
    if isinstance(fill_value, (int, float)):
        return edge_attr.new_full(size, fill_value)
    if isinstance(fill_value, Tensor):
        attr = fill_value.to(edge_attr.dtype)
        if edge_attr.dim() != attr.dim():
            attr = attr.unsqueeze(0)
        return attr.expand(size).contiguous()
    return scatter(edge_attr, edge_index[1], 0, num_nodes, fill_value)


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    """utils/loop.py:382."""
    N = num_nodes if num_nodes is not None else int(edge_index.max()) + 1
    loop = torch.arange(0, N, dtype=edge_index.dtype).view(1, -1).repeat(2, 1)
    if edge_attr is not None:
        edge_attr = torch.cat([edge_attr, _loop_attr(edge_index, edge_attr, N, fill_value)], 0)
    return torch.cat([edge_index, loop], dim=1), edge_attr


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    """utils/loop.py:585-657."""
    N = num_nodes if num_nodes is not None else int(edge_index.max()) + 1
    mask = edge_index[0] != edge_index[1]
    loop = torch.arange(0, N, dtype=edge_index.dtype).view(1, -1).repeat(2, 1)
    if edge_attr is not None:
        loop_attr = _loop_attr(edge_index, edge_attr, N, fill_value)
        inv = ~mask
        loop_attr[edge_index[0][inv]] = edge_attr[inv]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    return torch.cat([edge_index[:, mask], loop], dim=1), edge_attr


# ---- utils/_sort_edge_index.py, utils/_coalesce.py, utils/undirected.py ---------------------------
def _compound_key(edge_index: Tensor, num_nodes: int, sort_by_row: bool) -> Tensor:
    major, minor = (edge_index[0], edge_index[1]) if sort_by_row else (edge_index[1],
                                                                         edge_index[0])
    return major.long() * num_nodes + minor.long()


def sort_edge_index(edge_index: Tensor, edge_attr=None, num_nodes: Optional[int] = None,
                    sort_by_row: bool = True):
    """utils/_sort_edge_index.py:103-133: sort the key row*n+col (col*n+row), permute everything.
    Returns (edge_index, edge_attr) — attr a tensor, a list of tensors or None."""
    n = int(edge_index.max()) + 1 if num_nodes is None and edge_index.numel() else (num_nodes
                                                                                     or 0)
    perm = torch.sort(_compound_key(edge_index, n, sort_by_row), stable=True).indices
    if isinstance(edge_attr, (list, tuple)):
        return edge_index[:, perm], [a[perm] for a in edge_attr]
    return edge_index[:, perm], None if edge_attr is None else edge_attr[perm]


def coalesce(edge_index: Tensor, edge_attr=None, num_nodes: Optional[int] = None,
             reduce: str = 'sum', is_sorted: bool = False, sort_by_row: bool = True):
    """utils/_coalesce.py:131-193: sort by the compound key, keep the first edge of every run of
    equal keys, merge the attributes of a run with scatter(reduce)."""
    n = int(edge_index.max()) + 1 if num_nodes is None and edge_index.numel() else (num_nodes
                                                                                     or 0)
    key = _compound_key(edge_index, n, sort_by_row)
    attrs = list(edge_attr) if isinstance(edge_attr, (list, tuple)) else edge_attr
    if not is_sorted:
        key, perm = torch.sort(key, stable=True)
        edge_index = edge_index[:, perm]
        if isinstance(attrs, list):
            attrs = [a[perm] for a in attrs]
        elif attrs is not None:
            attrs = attrs[perm]
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    if bool(first.all()):
        return edge_index, attrs
    group = first.long().cumsum(0) - 1
    n_out = int(first.sum())
    edge_index = edge_index[:, first]
    if isinstance(attrs, list):
        attrs = [scatter(a, group, 0, n_out, reduce) for a in attrs]
    elif attrs is not None:
        attrs = scatter(attrs, group, 0, n_out, reduce)
    return edge_index, attrs


def to_undirected(edge_index: Tensor, edge_attr=None, num_nodes: Optional[int] = None,
                  reduce: str = 'add'):
    """utils/undirected.py:176-190: append the reversed edges (attributes repeated), coalesce."""
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if isinstance(edge_attr, (list, tuple)):
        edge_attr = [torch.cat([a, a], dim=0) for a in edge_attr]
    elif edge_attr is not None:
        edge_attr = torch.cat([edge_attr, edge_attr], dim=0)
    return coalesce(both, edge_attr, num_nodes, reduce)


def is_undirected(edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                  num_nodes: Optional[int] = None) -> bool:
    """utils/undirected.py:57-80: the row-sorted list must be the transpose of the column-sorted
    list, attributes included."""
    a_idx, a_attr = sort_edge_index(edge_index, edge_attr, num_nodes, True)
    b_idx, b_attr = sort_edge_index(edge_index, edge_attr, num_nodes, False)
    same = torch.equal(a_idx[0], b_idx[1]) and torch.equal(a_idx[1], b_idx[0])
    if same and edge_attr is not None:
        same = torch.equal(a_attr, b_attr)
    return same


# ---- layers ----------------------------------------------------------------------------------------
def gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False,
             add_self_loops_=True):
    """nn/conv/gcn_conv.py:94-113 (dense tensor branch, flow source_to_target)."""
    fill = 2. if improved else 1.
    if add_self_loops_:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill,
                                                           num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1))
    row, col = edge_index[0], edge_index[1]
    deg = scatter(edge_weight, col, 0, num_nodes, 'sum')
    dis = deg.pow_(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    return edge_index, dis[row] * edge_weight * dis[col]


def sage_conv(x, edge_index, w_l, b_l, w_r, aggr='mean', num_dst=None, x_dst=None):
    """nn/conv/sage_conv.py:120-144: ``lin_l(aggr_j x_j) + lin_r(x_i)``."""
    x_dst = x if x_dst is None else x_dst
    num_dst = x_dst.size(0) if num_dst is None else num_dst
    out = F.linear(propagate(x, edge_index, num_dst, aggr), w_l, b_l)
    if w_r is not None:
        out = out + F.linear(x_dst, w_r)
    return out


def graph_conv(x, edge_index, w_rel, b_rel, w_root, edge_weight=None, aggr='add'):
    """nn/conv/graph_conv.py:78-95: ``lin_rel(aggr_j e_ji x_j) + lin_root(x_i)``."""
    out = F.linear(propagate(x, edge_index, x.size(0), aggr, edge_weight), w_rel, b_rel)
    return out + F.linear(x, w_root)


def gcn_conv(x, edge_index, w, b, edge_weight=None, improved=False, add_self_loops_=True,
             normalize=True):
    """nn/conv/gcn_conv.py:227-268: normalise -> lin -> propagate(add) -> bias."""
    N = x.size(0)
    if normalize:
        edge_index, edge_weight = gcn_norm(edge_index, edge_weight, N, improved,
                                           add_self_loops_)
    out = propagate(F.linear(x, w), edge_index, N, 'sum', edge_weight)
    return out if b is None else out + b


def gat_conv(x, edge_index, w, att_src, att_dst, bias, heads, out_channels, concat=True,
             negative_slope=0.2, add_self_loops_=True, return_alpha=False):
    """nn/conv/gat_conv.py:254-409 (single weight ``lin``, no edge features, eval mode)."""
    H, C = heads, out_channels
    N = x.size(0)
    xs = F.linear(x, w).view(-1, H, C)
    a_src = (xs * att_src).sum(-1)
    a_dst = (xs * att_dst).sum(-1)
    if add_self_loops_:
        edge_index, _ = remove_self_loops(edge_index)
        edge_index, _ = add_self_loops(edge_index, num_nodes=N)
    alpha = a_src.index_select(0, edge_index[0]) + a_dst.index_select(0, edge_index[1])
    alpha = F.leaky_relu(alpha, negative_slope)
    alpha = softmax(alpha, edge_index[1], None, N)
    out = propagate(xs, edge_index, N, 'sum', alpha)
    out = out.view(-1, H * C) if concat else out.mean(dim=1)
    if bias is not None:
        out = out + bias
    return (out, edge_index, alpha) if return_alpha else out


def rgcn_conv(x, edge_index, edge_type, weight, root, bias, aggr='mean'):
    """nn/conv/rgcn_conv.py:164-282, default branch (no bases/blocks, float features): per
    relation masked propagate, ``h @ weight[r]``, summed; plus root and bias."""
    N = x.size(0)
    out = torch.zeros(N, weight.size(2))
    for r in range(weight.size(0)):
        tmp = edge_index[:, edge_type == r]
        out = out + propagate(x, tmp, N, aggr) @ weight[r]
    if root is not None:
        out = out + x @ root
    return out if bias is None else out + bias


def rgcn_conv_blocks(x, edge_index, edge_type, weight, root, bias, aggr='mean'):
    """nn/conv/rgcn_conv.py:207-220, block-diagonal branch: weight [R, B, in/B, out/B]."""
    N = x.size(0)
    R, B = weight.size(0), weight.size(1)
    out = torch.zeros(N, B * weight.size(3))
    for r in range(R):
        tmp = edge_index[:, edge_type == r]
        h = propagate(x, tmp, N, aggr).view(-1, B, weight.size(2))
        h = torch.einsum('abc,bcd->abd', h, weight[r])
        out = out + h.contiguous().view(N, -1)
    if root is not None:
        out = out + x @ root
    return out if bias is None else out + bias


def rgcn_conv_blocks_pairs(x, edge_index, edge_type, weight, root, bias, aggr='mean'):
    """:func:`rgcn_conv_blocks` (nn/conv/rgcn_conv.py:207-220) evaluated only on the rows that can
    be non-zero.  The reference multiplies, per relation r, ALL N rows of ``h = propagate_r(x)`` by
    ``weight[r]`` — N x R x B block products, 0.7 TFLOP per layer at the FB15k-237 shape, 474
    saved [N, F] tensors for the backward — although ``h`` is zero outside the destinations that
    relation r reaches.  Here the same sums are taken over the (relation, destination) PAIRS that
    exist: ``out[i] = sum_{r: (r,i) exists} aggr_{j in N_r(i)} x_j @ weight[r]``.  Identical math
    (a zero row times a matrix is a zero row), seconds instead of minutes at the timed shape of
    BASELINE config 5; pinned against the golden vectors generated by the reference itself and
    against :func:`rgcn_conv_blocks` in tests/test_oracle_golden.py.  ``weight``: [R, B, K, N]
    (or [R, K, N] for the dense branch, rgcn_conv.py:243-282)."""
    if weight.dim() == 3:
        weight = weight.unsqueeze(1)
    N = x.size(0)
    R, B, K, Nn = weight.shape
    key = edge_type.long() * N + edge_index[1].long()
    uniq, inv = torch.unique(key, return_inverse=True)   # ascending: grouped by relation
    S = uniq.numel()
    agg = scatter(x.index_select(0, edge_index[0]), inv, 0, S, 'sum' if aggr == 'add' else aggr)
    rel, dst = uniq // N, uniq % N
    ptr = index2ptr(rel, R).tolist()
    # (split, not slices: one concatenation in the backward instead of one [S, F] zero-filled
    # tensor per relation)
    chunks = torch.split(agg, [ptr[r + 1] - ptr[r] for r in range(R)])
    mats = weight.unbind(0)   # (likewise: one stack in the backward)
    parts = []
    for r in range(R):
        if chunks[r].size(0) == 0:
            continue
        h = chunks[r].reshape(-1, B, K)
        parts.append(torch.einsum('abc,bcd->abd', h, mats[r]).reshape(-1, B * Nn))
    out = x.new_zeros(N, B * Nn)
    if parts:
        out = out.index_add(0, dst, torch.cat(parts))
    if root is not None:
        out = out + x @ root
    return out if bias is None else out + bias


def rgcn_weight_from_bases(comp, bases, in_channels, out_channels):
    """nn/conv/rgcn_conv.py:203-205: weight[r] = sum_b comp[r, b] * bases[b]."""
    return (comp @ bases.view(bases.size(0), -1)).view(comp.size(0), in_channels, out_channels)


def rgcn_conv_index(node_index, edge_index, edge_type, weight, root, bias, aggr='mean',
                    num_dst=None, by_node_id=False):
    """Node-index ("featureless") inputs.  RGCNConv (nn/conv/rgcn_conv.py:262-268): per relation,
    propagate the embedding rows ``weight[r, node_index[j]]`` over the relation's edges and sum
    the relations.  FastRGCNConv (``by_node_id``, :357-359, :362-374): rows are looked up by the
    SOURCE NODE id, and a mean is a per-edge 1 / |N_r(i)| weight followed by one scatter-add."""
    n_dst = node_index.size(0) if num_dst is None else num_dst
    out = torch.zeros(n_dst, weight.size(2))
    for r in range(weight.size(0)):
        tmp = edge_index[:, edge_type == r]
        look = tmp[0] if by_node_id else node_index[tmp[0]]
        out = out + scatter(weight[r, look], tmp[1], 0, n_dst, 'sum' if aggr == 'add' else aggr)
    if root is not None:
        out = out + root[node_index]
    return out if bias is None else out + bias


# ---- models (nn/models/basic_gnn.py:178-274) ------------------------------------------------------
def graphsage(x, edge_index, params: List[Tuple[Tensor, Tensor, Tensor]], aggr='mean'):
    """GraphSAGE: SAGEConv layers with ReLU between them, none after the last."""
    for i, (w_l, b_l, w_r) in enumerate(params):
        x = sage_conv(x, edge_index, w_l, b_l, w_r, aggr)
        if i < len(params) - 1:
            x = x.relu()
    return x


def gcn(x, edge_index, params: List[Tuple[Tensor, Tensor]], edge_weight=None):
    for i, (w, b) in enumerate(params):
        x = gcn_conv(x, edge_index, w, b, edge_weight)
        if i < len(params) - 1:
            x = x.relu()
    return x


def gat(x, edge_index, params, heads: int):
    """GAT: hidden layers concat heads, explicit output layer averages them."""
    for i, (w, a_s, a_d, b) in enumerate(params):
        last = i == len(params) - 1
        C = a_s.size(-1)
        x = gat_conv(x, edge_index, w, a_s, a_d, b, heads, C, concat=not last)
        if not last:
            x = x.relu()
    return x

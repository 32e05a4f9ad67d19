"""``sort_edge_index`` / ``coalesce`` / ``to_undirected`` / ``is_undirected`` on the device
 — the preprocessing steps either side of the hot path.

Same contracts as torch_geometric/utils/_sort_edge_index.py:61-133, utils/_coalesce.py:72-193 and
utils/undirected.py:37-190: the compound key ``major * num_nodes + minor`` is sorted with the HIP
radix sort (stable, so equal keys keep their input order — one of the orders the reference's
unstable sort may produce), decoded back to ``(row, col)`` without a second gather, and — for
``coalesce`` — compacted to one entry per run of equal keys, the edge attributes being merged by
ONE scatter over the original (unsorted) attribute rows."""
from typing import List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from .. import _native
from ._scatter import scatter
from .num_nodes import maybe_num_nodes

MISSING = '???'
_MAX_INT64 = torch.iinfo(torch.int64).max
Attr = Union[Optional[Tensor], List[Tensor], str]


def _pair(edge_index) -> Tuple[Tensor, Tensor]:
    if isinstance(edge_index, Tensor):
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"'edge_index' needs the shape [2, num_edges] "
                             f"(got {list(edge_index.size())})")
        return edge_index[0], edge_index[1]
    if isinstance(edge_index, tuple) and len(edge_index) == 2:
        return edge_index
    raise NotImplementedError


def _like_input(edge_index, out: Tensor):
    return (out[0], out[1]) if isinstance(edge_index, tuple) else out


def _take(edge_attr: Attr, perm: Tensor) -> Attr:
    if isinstance(edge_attr, Tensor):
        return edge_attr[perm]
    if isinstance(edge_attr, (list, tuple)):
        return [e[perm] for e in edge_attr]
    return edge_attr


def _result(edge_index, edge_attr: Attr):
    if edge_attr is None or isinstance(edge_attr, (Tensor, list, tuple)):
        return edge_index, edge_attr
    return edge_index  # MISSING: the caller did not pass attributes at all


def sort_edge_index(edge_index, edge_attr: Attr = MISSING, num_nodes: Optional[int] = None,
                    sort_by_row: bool = True):
    r"""Sorts the edges by row (then column), or by column (then row) with
    ``sort_by_row=False``; attributes (a tensor or a list of tensors) follow their edges.  Returns
    ``edge_index`` alone when no ``edge_attr`` argument was passed, else a tuple."""
    row, col = _pair(edge_index)
    n = maybe_num_nodes(torch.stack([row, col]) if isinstance(edge_index, tuple) else edge_index,
                        num_nodes)
    if n * n > _MAX_INT64:
        raise ValueError("'sort_edge_index' would overflow the compound int64 sort key")
    if row.numel() == 0:
        return _result(edge_index, edge_attr)
    key = _native.edge_key(row, col, n, sort_by_row)
    key, perm = _native.index_sort(key, n * n)
    out, _ = _native.edge_unkey(key, n, sort_by_row, row.dtype)
    return _result(_like_input(edge_index, out), _take(edge_attr, perm))


def coalesce(edge_index, edge_attr: Attr = MISSING, num_nodes: Optional[int] = None,
             reduce: str = 'sum', is_sorted: bool = False, sort_by_row: bool = True):
    r"""Sorts the edges and removes duplicates, merging duplicate attributes with ``reduce``
    (``sum`` / ``add`` / ``mean`` / ``min`` / ``max`` / ``mul`` / ``any``).  ``is_sorted=True``
    promises the input is already in the requested order."""
    row, col = _pair(edge_index)
    n = maybe_num_nodes(torch.stack([row, col]) if isinstance(edge_index, tuple) else edge_index,
                        num_nodes)
    if n * n > _MAX_INT64:
        raise ValueError("'coalesce' will result in an overflow")
    E = row.numel()
    if E == 0:
        return _result(edge_index, edge_attr)
    key = _native.edge_key(row, col, n, sort_by_row)
    perm = None
    if not is_sorted:
        key, perm = _native.index_sort(key, n * n)
    scan = _native.cumsum(_native.run_flags(key))
    n_unique = int(scan[-1])  # host sync, where the reference has `mask.all()` / mask indexing
    if n_unique == E:  # nothing to merge: only the order changes
        if perm is None:
            return _result(edge_index, edge_attr)
        out, _ = _native.edge_unkey(key, n, sort_by_row, row.dtype)
        return _result(_like_input(edge_index, out), _take(edge_attr, perm))
    has_attr = isinstance(edge_attr, Tensor) or (isinstance(edge_attr, (list, tuple))
                                                 and len(edge_attr) > 0)
    out, gid = _native.edge_unkey(key, n, sort_by_row, row.dtype, scan=scan, perm=perm,
                                  n_out=n_unique, want_groups=has_attr)
    out = _like_input(edge_index, out)
    if isinstance(edge_attr, Tensor):
        return out, scatter(edge_attr, gid, 0, n_unique, reduce)
    if isinstance(edge_attr, (list, tuple)):
        return out, [scatter(e, gid, 0, n_unique, reduce) for e in edge_attr]
    return _result(out, edge_attr)


def to_undirected(edge_index: Tensor, edge_attr: Attr = MISSING, num_nodes: Optional[int] = None,
                  reduce: str = 'add'):
    r"""Adds the reverse of every edge and coalesces (undirected.py:135-190)."""
    if isinstance(edge_attr, int):  # legacy call: to_undirected(edge_index, num_nodes)
        num_nodes, edge_attr = edge_attr, MISSING
    row, col = _pair(edge_index)
    both = torch.stack([torch.cat([row, col]), torch.cat([col, row])])
    if isinstance(edge_attr, Tensor):
        edge_attr = torch.cat([edge_attr, edge_attr], dim=0)
    elif isinstance(edge_attr, (list, tuple)):
        edge_attr = [torch.cat([e, e], dim=0) for e in edge_attr]
    return coalesce(both, edge_attr, num_nodes, reduce)


def is_undirected(edge_index: Tensor, edge_attr: Union[Optional[Tensor], Sequence[Tensor]] = None,
                  num_nodes: Optional[int] = None) -> bool:
    r"""True if for every edge (with its attributes) the reverse edge is present with the same
    attributes (undirected.py:37-80): the row-sorted and the column-sorted edge lists must be
    transposes of each other."""
    n = maybe_num_nodes(edge_index, num_nodes)
    attrs: List[Tensor] = []
    if isinstance(edge_attr, Tensor):
        attrs = [edge_attr]
    elif isinstance(edge_attr, (list, tuple)):
        attrs = list(edge_attr)
    by_row, attrs_r = sort_edge_index(edge_index, attrs, n, sort_by_row=True)
    by_col, attrs_c = sort_edge_index(edge_index, attrs, n, sort_by_row=False)
    if not (torch.equal(by_row[0], by_col[1]) and torch.equal(by_row[1], by_col[0])):
        return False
    return all(torch.equal(a, b) for a, b in zip(attrs_r, attrs_c))

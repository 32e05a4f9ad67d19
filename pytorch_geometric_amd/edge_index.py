"""Graph handle: COO ``edge_index`` plus cached CSR (sorted by destination), CSC (sorted by
source), the two stable permutations and the hub plans the SpMM kernels need.

Mirrors the cache of the reference's ``EdgeIndex`` (torch_geometric/edge_index.py:589-696:
``get_indptr`` / ``_sort_by_transpose`` / ``get_csr`` / ``get_csc`` / ``fill_cache_``).  Like the
reference's (edge_index.py:173 ``class EdgeIndex(Tensor)``) the handle IS the ``[2, E]`` tensor: a
``torch.Tensor`` subclass sharing the storage of the edge list it was built from, so it can be
indexed, passed to ``torch`` functions and handed to code that expects a plain ``edge_index``.
Unlike the reference it does not dispatch: ``__torch_function__`` is disabled, every ``torch``
operation on a handle returns a plain tensor and the cache stays on the Python object (the
sorted forms are used by this package's kernels, never rebuilt by an intercepted ``aten`` op).
Message flow is ``source_to_target``: ``edge_index[0]`` = source j, ``edge_index[1]`` =
destination i (collect.jinja:31,67-68).

All integer outputs are bit-exact with ``torch.sort(stable=True)`` +
``torch._convert_indices_from_coo_to_csr`` (tests/test_gpu_graph.py).
"""
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _native


class CSR:
    """One sorted orientation: ``ptr`` over the sorted rows, ``idx`` = the other endpoint per slot,
    ``perm`` = slot -> position in the COO edge list (stable), ``hub`` = split plan for long rows."""

    def __init__(self, ptr: Tensor, idx: Tensor, perm: Tensor, n_rows: int, n_cols: int):
        self.ptr, self.idx, self.perm = ptr, idx, perm
        self.n_rows, self.n_cols = n_rows, n_cols
        self._hub = None
        self._inv_deg = None

    @property
    def hub(self):
        if self._hub is None:
            self._hub = _native.hub_plan(self.ptr)
        return self._hub

    @property
    def nnz(self) -> int:
        return self.idx.numel()

    def degree(self) -> Tensor:
        return self.ptr[1:] - self.ptr[:-1]

    def inv_degree(self) -> Tensor:
        """1 / clamp(deg, 1) as fp32 — the mean normaliser (utils/_scatter.py:72-80)."""
        if self._inv_deg is None:
            self._inv_deg = 1.0 / self.degree().clamp(min=1).to(torch.float32)
        return self._inv_deg


def build_csr(key: Tensor, other: Tensor, n_rows: int, n_cols: int,
              is_sorted: bool = False) -> CSR:
    """Stable sort of the COO list by ``key`` (edge_index.py:605-623 ``_sort_by_transpose``)."""
    if is_sorted:
        perm = torch.arange(key.numel(), dtype=key.dtype, device=key.device)
        ptr = _native.index2ptr(key, n_rows)
        return CSR(ptr, other.contiguous(), perm, n_rows, n_cols)
    sorted_key, perm64 = _native.index_sort(key, max_value=max(n_rows - 1, 0))
    idx = _native.permute_index(other, perm64)
    ptr = _native.index2ptr(sorted_key, n_rows)
    perm = _native.cast_index(perm64, key.dtype)
    return CSR(ptr, idx, perm, n_rows, n_cols)


def _plain(edge_index: Tensor) -> Tensor:
    return edge_index.edge_index if isinstance(edge_index, EdgeIndex) else edge_index


class EdgeIndex(Tensor):
    r"""COO edge list of a (bipartite) graph with lazily built, cached sorted forms.

    Args:
        edge_index: ``[2, E]`` int32/int64 device tensor (row 0 = source, row 1 = destination).
        sparse_size: ``(num_src_nodes, num_dst_nodes)``; inferred (one host sync) when ``None``,
            like ``maybe_num_nodes`` in the reference.
        sort_order: ``'row'`` if ``edge_index[0]`` is already sorted, ``'col'`` if ``edge_index[1]``
            is, else ``None`` (same vocabulary as the reference's ``EdgeIndex``).
        validate: range-check the indices against ``sparse_size`` once (two host syncs) and check
            a claimed ``sort_order`` (one more); ``'range'``: the range check only (the caller
            vouches for the order, as the reference's ``EdgeIndex`` does for its own flag).
    """

    # every torch operation on a handle sees (and returns) a plain tensor
    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def __new__(cls, edge_index: Tensor, sparse_size: Optional[Tuple[int, int]] = None,
                sort_order: Optional[str] = None, validate: bool = True):
        if not isinstance(edge_index, Tensor):
            raise ValueError(f"'edge_index' must be a Tensor (got {type(edge_index)})")
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"'edge_index' needs to be of shape [2, num_edges] "
                             f"(got {list(edge_index.shape)})")
        if edge_index.dtype not in (torch.int32, torch.int64):
            raise ValueError(f"'edge_index' holds an unsupported data type "
                             f"(got '{edge_index.dtype}', expected int32 or int64)")
        if sort_order not in (None, 'row', 'col'):
            raise ValueError(f"invalid sort_order '{sort_order}'")
        # shares storage, strides and version counter with `edge_index` (like nn.Parameter)
        return Tensor._make_subclass(cls, _plain(edge_index), False)

    def __init__(self, edge_index: Tensor, sparse_size: Optional[Tuple[int, int]] = None,
                 sort_order: Optional[str] = None, validate: bool = True):
        edge_index = _plain(edge_index)
        # the plain tensor the handle was built from: what kernels, caches and `narrow` use
        self.edge_index = edge_index
        if sparse_size is None or sparse_size[0] is None or sparse_size[1] is None:
            n = 0
            if edge_index.numel() > 0:
                n = _native.index_minmax(edge_index.reshape(-1))[1] + 1
            given = sparse_size or (None, None)
            sparse_size = (n if given[0] is None else given[0],
                           n if given[1] is None else given[1])
        self.sparse_size = (int(sparse_size[0]), int(sparse_size[1]))
        if validate in (True, 'range') and edge_index.numel() > 0:
            # the reference raises IndexError from index_select (message_passing.py:269-290);
            # the fused kernels never bounds-check, so the handle does it once, up front
            for row, n in ((0, self.sparse_size[0]), (1, self.sparse_size[1])):
                lo, hi = _native.index_minmax(edge_index[row])
                if lo < 0:
                    raise IndexError(
                        f"Found negative indices in 'edge_index' (got {lo}). Please ensure that "
                        f"all indices in 'edge_index' point to valid indices in the interval "
                        f"[0, {n}) in your node feature matrix and try again.")
                if hi >= n:
                    raise IndexError(
                        f"Found indices in 'edge_index' that are larger than {n - 1} (got "
                        f"{hi}). Please ensure that all indices in 'edge_index' point to valid "
                        f"indices in the interval [0, {n}) in your node feature matrix and try "
                        f"again.")
        if validate is True and sort_order is not None and edge_index.size(1) > 1:
            key = edge_index[0] if sort_order == 'row' else edge_index[1]
            if not bool((key[1:] >= key[:-1]).all()):
                raise ValueError(f"'edge_index' is not sorted by {sort_order} although "
                                 f"sort_order='{sort_order}' was given")
        self.sort_order = sort_order
        self._csr: Optional[CSR] = None   # sorted by destination (aggregation / forward)
        self._csc: Optional[CSR] = None   # sorted by source (transposed / backward)
        self._slot_map = None
        self._flipped: Optional['EdgeIndex'] = None
        self._derived = {}
        # True: the backward of an aggregation runs the edge-parallel atomic kernel on the COO
        # list instead of building the source-sorted form (graphs used once: sampled batches)
        self.atomic_backward = False

    @classmethod
    def from_csr(cls, rowptr: Tensor, col: Tensor, sparse_size: Tuple[int, int]) -> 'EdgeIndex':
        """Handle for an ``adj_t`` already in CSR form (rows = destinations, ``col`` = sources;
        ``sparse_size = (num_src, num_dst)``): no sort is needed for the forward."""
        dst = _native.ptr2index(rowptr, col.numel())
        ei = torch.stack([col, dst])
        self = cls(ei, sparse_size, sort_order='col', validate=False)
        perm = torch.arange(col.numel(), dtype=col.dtype, device=col.device)
        self._csr = CSR(rowptr.contiguous(), col.contiguous(), perm, sparse_size[1],
                        sparse_size[0])
        return self

    @classmethod
    def from_sorted_batch(cls, edge_index: Tensor, num_nodes: int,
                          max_in_degree: Optional[int] = None) -> 'EdgeIndex':
        """Handle for a sampled mini-batch whose edges are already ordered by destination (the
        sampler emits them hop by hop and, inside a hop, by destination): no sort, no range
        check, no host sync; the backward uses the atomic COO kernel (the batch is used once)."""
        self = cls(edge_index, (num_nodes, num_nodes), sort_order='col', validate=False)
        self.atomic_backward = True
        csr = self.by_dst()
        if max_in_degree is not None and max_in_degree <= _native.HUB_THRESHOLD:
            csr._hub = (None, None, 0, 0)  # fan-out bounded: no row needs splitting
        return self

    def record_stream(self, stream) -> None:
        """Tell the caching allocator that `stream` uses the tensors of this handle (a handle
        built on a side stream — the prefetching loader — and consumed on another).  Covers
        ``Tensor.record_stream`` of the handle itself: it shares ``edge_index``'s storage."""
        tensors = [self.edge_index]
        for csr in (self._csr, self._csc):
            if csr is not None:
                tensors += [csr.ptr, csr.idx, csr.perm, csr._inv_deg]
                if csr._hub is not None:
                    tensors += [csr._hub[0], csr._hub[1]]
        for t in tensors:
            if isinstance(t, Tensor) and t.is_cuda:
                t.record_stream(stream)

    def trim(self, num_nodes: int, num_edges: int) -> 'EdgeIndex':
        """Prefix handle for ``trim_to_layer`` on hop-ordered batches: keeps the first
        ``num_edges`` edges and the first ``num_nodes`` nodes.  Destination-sorted handles are
        trimmed without re-sorting: ``ptr`` is clipped at ``num_edges``."""
        ei = self.edge_index.narrow(1, 0, num_edges)
        if self.sort_order != 'col' or self._csr is None:
            out = EdgeIndex(ei, (num_nodes, num_nodes), sort_order=self.sort_order,
                            validate=False)
            out.atomic_backward = self.atomic_backward
            return out
        out = EdgeIndex(ei, (num_nodes, num_nodes), sort_order='col', validate=False)
        out.atomic_backward = self.atomic_backward
        full = self._csr
        ptr = full.ptr.narrow(0, 0, num_nodes + 1).clamp(max=num_edges)
        csr = CSR(ptr, full.idx.narrow(0, 0, num_edges), full.perm.narrow(0, 0, num_edges),
                  num_nodes, num_nodes)
        if full._hub is not None and full._hub[2] == 0:
            csr._hub = (None, None, 0, 0)
        out._csr = csr
        return out

    # -- accessors -------------------------------------------------------------------------------
    @property
    def num_edges(self) -> int:
        return self.edge_index.size(1)

    @property
    def num_src_nodes(self) -> int:
        return self.sparse_size[0]

    @property
    def num_dst_nodes(self) -> int:
        return self.sparse_size[1]

    def by_dst(self) -> CSR:
        """ptr over destinations, idx = sources (``get_csc`` in the reference's row/col naming:
        the form ``message_and_aggregate`` consumes, edge_index.py:646-663)."""
        if self._csr is None:
            src, dst = self.edge_index[0], self.edge_index[1]
            self._csr = build_csr(dst, src, self.num_dst_nodes, self.num_src_nodes,
                                  is_sorted=self.sort_order == 'col')
        return self._csr

    def by_src(self) -> CSR:
        """ptr over sources, idx = destinations (the transposed handle for the backward,
        edge_index.py:1849-1900)."""
        if self._csc is None:
            src, dst = self.edge_index[0], self.edge_index[1]
            self._csc = build_csr(src, dst, self.num_src_nodes, self.num_dst_nodes,
                                  is_sorted=self.sort_order == 'row')
        return self._csc

    # -- the reference's accessor names (torch_geometric/edge_index.py:626-663, 727-776, 970-1026)
    def get_csr(self):
        """``((rowptr, col), perm)``: pointer over ``edge_index[0]``, ``col = edge_index[1]`` sorted
        by row (stable), ``perm`` = sorted position -> COO position."""
        c = self.by_src()
        return (c.ptr, c.idx), c.perm

    def get_csc(self):
        """``((colptr, row), perm)``: pointer over ``edge_index[1]``, ``row = edge_index[0]``
        sorted by column (stable)."""
        c = self.by_dst()
        return (c.ptr, c.idx), c.perm

    def sort_by(self, sort_order: str):
        """``(sorted EdgeIndex, perm)`` with ``edge_index[:, perm]`` ordered by ``'row'``
        (``edge_index[0]``) or ``'col'`` (``edge_index[1]``); stable, bit-identical to
        ``torch.sort(stable=True)`` on the key row."""
        if sort_order not in ('row', 'col'):
            raise ValueError(f"invalid sort_order '{sort_order}'")
        if self.sort_order == sort_order:
            return self, None
        c = self.by_src() if sort_order == 'row' else self.by_dst()
        keys = _native.ptr2index(c.ptr, c.idx.numel())
        ei = torch.stack([keys, c.idx]) if sort_order == 'row' else torch.stack([c.idx, keys])
        out = EdgeIndex(ei, self.sparse_size, sort_order=sort_order, validate=False)
        return out, c.perm

    def matmul(self, other: Tensor, input_value: Optional[Tensor] = None, reduce: str = 'sum',
               transpose: bool = False) -> Tensor:
        r"""Sparse-dense product with the handle as the matrix ``A[edge_index[0], edge_index[1]]``
        (``EdgeIndex.matmul``, edge_index.py:970-1026): ``A @ other`` reduces ``other[col]`` onto
        the rows; ``transpose=True`` gives ``A^T @ other``, i.e. what ``propagate`` computes
        (messages from ``edge_index[0]`` onto ``edge_index[1]``)."""
        from ._functions import SpmmFunction
        reduce = 'sum' if reduce == 'add' else reduce
        if reduce not in ('sum', 'mean', 'min', 'max'):
            raise ValueError(f"`reduce` argument '{reduce}' not supported")
        if transpose:
            return SpmmFunction.apply(other, input_value, self, reduce, 'coo')
        if self._flipped is None:
            flipped = EdgeIndex(self.edge_index.flip(0).contiguous(),
                                (self.sparse_size[1], self.sparse_size[0]), validate=False)
            flipped._csr, flipped._csc = self._csc, self._csr  # the sorted forms swap roles
            self._flipped = flipped
        return SpmmFunction.apply(other, input_value, self._flipped, reduce, 'coo')

    def derived(self, key, build):
        """Per-handle cache of graphs DERIVED from this one by a layer (``gcn_norm``'s
        self-looped, normalised edge list; GAT's re-self-looped edge list): ``build()`` runs once
        per ``key``; the derived handle keeps its own sorted forms."""
        hit = self._derived.get(key)
        if hit is None:
            if len(self._derived) >= 4:
                self._derived.pop(next(iter(self._derived)))
            hit = self._derived[key] = build()
        return hit

    def fill_cache_(self) -> 'EdgeIndex':
        self.by_dst().hub
        self.by_src().hub
        return self

    def src_slot_to_dst_slot(self) -> Tensor:
        """For every slot of the by-source form, the slot of the same edge in the by-destination
        form (needed when per-edge values live in by-destination slot order, e.g. GAT's alpha)."""
        if self._slot_map is None:
            fwd, bwd = self.by_dst(), self.by_src()
            inv = torch.empty_like(fwd.perm)
            inv[fwd.perm.long()] = torch.arange(fwd.perm.numel(), dtype=fwd.perm.dtype,
                                                device=fwd.perm.device)
            self._slot_map = inv[bwd.perm.long()].contiguous()
        return self._slot_map

    # -- tensor protocol -------------------------------------------------------------------------
    def as_tensor(self) -> Tensor:
        """The plain ``[2, E]`` tensor (``EdgeIndex.as_tensor``, edge_index.py:904-909)."""
        return self.edge_index

    def to(self, *args, **kwargs):
        """Device / dtype moves keep the handle (sizes and sort order; the sorted forms are
        rebuilt lazily on the new device); a move to a non-index dtype returns a plain tensor."""
        out = self.edge_index.to(*args, **kwargs)
        if out is self.edge_index:
            return self
        if out.dtype not in (torch.int32, torch.int64):
            return out
        moved = EdgeIndex(out, self.sparse_size, sort_order=self.sort_order, validate=False)
        moved.atomic_backward = self.atomic_backward
        return moved

    def cuda(self, *args, **kwargs):
        return self.to(torch.device('cuda', *args), **kwargs) if args else self.to('cuda', **kwargs)

    def cpu(self):
        return self.to('cpu')

    def __deepcopy__(self, memo):
        out = EdgeIndex(self.edge_index.clone(), self.sparse_size, sort_order=self.sort_order,
                        validate=False)
        out.atomic_backward = self.atomic_backward
        memo[id(self)] = out
        return out

    def __reduce_ex__(self, protocol):
        return (_rebuild, (self.edge_index, self.sparse_size, self.sort_order,
                           self.atomic_backward))

    def __repr__(self) -> str:
        return (f'EdgeIndex(num_edges={self.num_edges}, sparse_size={self.sparse_size}, '
                f'sort_order={self.sort_order})')


def _rebuild(edge_index, sparse_size, sort_order, atomic_backward):
    out = EdgeIndex(edge_index, sparse_size, sort_order=sort_order, validate=False)
    out.atomic_backward = atomic_backward
    return out


# ---- cache for raw ``edge_index`` tensors ------------------------------------------------------
# nn.conv layers receive a plain tensor every call; sorting it once per (tensor, version, size)
# turns 3 layers x N epochs of scatter into CSR SpMM.
_cache = {}
_cache_enabled = True
MAX_CACHE_ENTRIES = 8


def set_cache_enabled(flag: bool):
    global _cache_enabled
    _cache_enabled = bool(flag)
    if not flag:
        _cache.clear()


def clear_cache():
    _cache.clear()


def as_edge_index(edge_index, num_src: Optional[int] = None, num_dst: Optional[int] = None,
                  flip: bool = False) -> EdgeIndex:
    """Handle for a raw ``[2, E]`` tensor, cached per (tensor identity, version, sizes, flip).
    ``flip=True`` swaps the two rows first (``flow='target_to_source'``)."""
    return adopt_sorted(edge_index, num_src, num_dst, flip, None)[0]


def adopt_sorted(edge_index, num_src: Optional[int], num_dst: Optional[int], flip: bool,
                 sort_order: Optional[str]) -> Tuple[EdgeIndex, bool]:
    """``as_edge_index`` for a tensor whose owner vouches for ``sort_order`` (of the handle's rows,
    i.e. AFTER the flip) — the data of a reference ``EdgeIndex``: the indices are still
    range-checked once, the order is taken on trust.  Returns ``(handle, fresh)``; ``fresh`` says
    the handle was built by this call (the caller may seed its sorted forms)."""
    if isinstance(edge_index, EdgeIndex):
        if flip:
            raise ValueError("an EdgeIndex handle is always 'source_to_target'; pass a raw "
                             "tensor to use flow='target_to_source'")
        return edge_index, False
    if not isinstance(edge_index, Tensor):
        raise ValueError(f"'edge_index' must be a Tensor or EdgeIndex (got {type(edge_index)})")

    def make():
        ei = edge_index.flip(0).contiguous() if flip else edge_index
        return EdgeIndex(ei, (num_src, num_dst), sort_order=sort_order,
                         validate=True if sort_order is None else 'range')

    if not _cache_enabled:
        return make(), True
    key = (id(edge_index), flip)
    hit = _cache.get(key)
    if hit is not None:
        ref, version, size, handle = hit
        if (ref() is edge_index and version == edge_index._version
                and size == (num_src, num_dst)):
            return handle, False
    handle = make()
    if len(_cache) >= MAX_CACHE_ENTRIES:
        _cache.pop(next(iter(_cache)))

    def _drop(_, key=key):
        _cache.pop(key, None)

    _cache[key] = (weakref.ref(edge_index, _drop), edge_index._version, (num_src, num_dst), handle)
    return handle, True

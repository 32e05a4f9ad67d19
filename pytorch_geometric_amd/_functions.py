"""Autograd glue: each ``torch.autograd.Function`` pairs one forward HIP kernel with the HIP kernels
of its backward.  Formulas follow the reference's CPU path (see the docstrings for file:line)."""
import collections
import os
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import math

from . import _native
from .edge_index import EdgeIndex


def _shaped(out2d: Tensor, shape) -> Tensor:
    """The freshly allocated 2-D kernel output under its public shape WITHOUT making it a view:
    autograd forbids in-place edits of views returned by a custom Function, and the reference does
    edit such outputs in place (``deg.pow_(-0.5)`` on a ``scatter`` result, gcn_conv.py:109)."""
    shape = tuple(shape)
    if tuple(out2d.shape) == shape:
        return out2d
    assert out2d.is_contiguous() and out2d._base is None
    return out2d.new_empty(0).set_(out2d.untyped_storage(), out2d.storage_offset(), shape)


def _rows(t: Tensor) -> Tensor:
    """[n, ...] -> [n, prod(...)] (also for empty tensors, where reshape(n, -1) is ambiguous)."""
    return t.reshape(t.size(0), math.prod(t.shape[1:]))


class SpmmFunction(Function):
    r"""``out[i] = reduce_{(j -> i)} w_e * x[j]`` on an :class:`EdgeIndex` handle.

    Forward = fused gather/message/reduce of ``MessagePassing.propagate``
    (nn/conv/message_passing.py:421-563; utils/_spmm.py:12-136).  Backward w.r.t. ``x`` = the same
    kernel on the transposed handle (edge_index.py:1849-1900); w.r.t. ``w`` = SDDMM
    (edge_index.py:1903-1922).  ``w_order``: 'coo' (``w[e]`` follows ``edge_index`` order) or
    'slot' (``w`` follows the by-destination slot order, e.g. GAT's alpha).
    """

    @staticmethod
    def forward(ctx, x: Tensor, w: Optional[Tensor], graph: EdgeIndex, reduce: str,
                w_order: str):
        fwd = graph.by_dst()
        if x.size(0) != graph.num_src_nodes:
            raise ValueError(f"'x' has {x.size(0)} rows but the graph has "
                             f"{graph.num_src_nodes} source nodes")
        ctx.graph, ctx.reduce, ctx.w_order = graph, reduce, w_order
        ctx.x_shape = x.shape
        x2 = _rows(x)
        if reduce in ('min', 'max'):
            if w is not None:
                raise NotImplementedError("edge weights are not supported for min/max")
            # the forward also leaves, per output, WHICH slot attained the extremum (int32) or a
            # "split" mark: the backward then needs no edge pass for the unique extrema
            save = ctx.needs_input_grad[0] and not torch.are_deterministic_algorithms_enabled()
            if save:
                out, arg32 = _native.spmm_csr(fwd.ptr, fwd.idx, x2, reduce, n_rows=fwd.n_rows,
                                              hub=fwd.hub, save_arg32=True)
            else:
                out, arg32 = _native.spmm_csr(fwd.ptr, fwd.idx, x2, reduce, n_rows=fwd.n_rows,
                                              hub=fwd.hub), None
            # (the RETURNED tensor is saved, not its 2-D alias: autograd's version check then sees an
            # in-place edit of the result between forward and backward)
            res = _shaped(out, (fwd.n_rows, *x.shape[1:]))
            ctx.save_for_backward(x2, res, arg32)
            return res
        else:
            eid = fwd.perm if (w is not None and w_order == 'coo') else None
            out = _native.spmm_csr(fwd.ptr, fwd.idx, x2, reduce, n_rows=fwd.n_rows, eid=eid, w=w,
                                   hub=fwd.hub)
            need_x = w is not None and ctx.needs_input_grad[1]
            ctx.save_for_backward(x2 if need_x else None, w)
        return _shaped(out, (fwd.n_rows, *x.shape[1:]))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        graph, reduce = ctx.graph, ctx.reduce
        g2 = _rows(grad_out)
        grad_x = grad_w = None
        if reduce in ('min', 'max'):
            x2, out, arg32 = ctx.saved_tensors
            out = _rows(out)
            if ctx.needs_input_grad[0]:
                fwd = graph.by_dst()
                if torch.are_deterministic_algorithms_enabled():
                    # source-driven, no atomics: fixed summation order, ~3.5x slower
                    bwd = graph.by_src()
                    ntie = _native.spmm_tie_count(fwd.ptr, fwd.idx, x2, out, count_self=True)
                    grad_x = _native.spmm_minmax_backward(bwd.ptr, bwd.idx, x2, out, g2, ntie)
                else:
                    grad_x = None
                    if arg32 is not None and not graph.atomic_backward:
                        # cached graph: winners as per-edge bit masks + a source-driven sum over
                        # the transposed CSR instead of N x F scattered atomics
                        grad_x = _native.spmm_minmax_backward_src(
                            fwd, graph.by_src(), graph.src_slot_to_dst_slot(), x2, out, g2, arg32)
                    if grad_x is None:
                        grad_x = _native.spmm_minmax_backward_dst(fwd.ptr, fwd.idx, x2, out, g2,
                                                                  graph.num_src_nodes,
                                                                  arg32=arg32)
                grad_x = grad_x.view(ctx.x_shape)
            return grad_x, None, None, None, None
        x2, w = ctx.saved_tensors
        fwd = graph.by_dst()
        if reduce == 'mean' and w is not None:
            g2 = g2 * fwd.inv_degree().view(-1, 1)
        if ctx.needs_input_grad[0]:
            scale = fwd.inv_degree() if (reduce == 'mean' and w is None) else None
            if graph.atomic_backward and (w is None or w.dim() == 1 or w.size(1) == 1):
                # graph used once (sampled batch): edge-parallel atomics on the COO list
                # instead of sorting by source (one weight per edge: also a single-head GAT)
                w_coo = w if w is None else w.reshape(-1)
                if w is not None and ctx.w_order == 'slot':
                    w_coo = torch.empty_like(w_coo)
                    w_coo[fwd.perm.long()] = w.reshape(-1)
                ei = graph.edge_index
                grad_x = _native.gather_scatter_add(g2, ei[1], ei[0], graph.num_src_nodes,
                                                    scale=scale, w=w_coo)
            else:
                bwd = graph.by_src()
                eid = None
                if w is not None:
                    eid = bwd.perm if ctx.w_order == 'coo' else graph.src_slot_to_dst_slot()
                grad_x = _native.spmm_csr(bwd.ptr, bwd.idx, g2, 'sum', n_rows=bwd.n_rows,
                                          eid=eid, w=w, src_scale=scale, hub=bwd.hub)
            grad_x = grad_x.view(ctx.x_shape)
        if w is not None and ctx.needs_input_grad[1]:
            eid = fwd.perm if ctx.w_order == 'coo' else None
            heads = 1 if w.dim() == 1 else w.size(1)
            grad_w = _native.sddmm_csr(fwd.ptr, fwd.idx, eid, g2, x2, fwd.nnz, heads)
            grad_w = grad_w.view(w.shape)
        return grad_x, grad_w, None, None, None


# ---- large unsorted scatters: sort once, then a segment reduction ---------------------------------
# The atomic scatter kernels move a cache line per 4-byte atomic: at the products shape
# (E = 61.9 M rows of 256 floats) `scatter(msg, edge_index[1])` takes 206 ms = 0.04 of the HBM peak,
# 311 ms for max (profiles/r04_unfused_propagate.md).  One stable radix sort of the index (a few ms,
# cached per index tensor) turns the same call into a gather-SpMM over the sorted groups
# (`col` = the sort permutation): 11.5 ms, deterministic.  Small inputs keep the atomics.
#
# Who takes the sorted route:
#   * an index of >= SORTED_SCATTER_ALWAYS_ROWS entries always: the sort costs a small fraction of
#     what the atomics cost there, even for an index tensor that is never seen again;
#   * a smaller one (>= SORTED_SCATTER_MIN_ROWS) only when the same tensor (identity + version) has
#     been seen before, or when it is a row of an `EdgeIndex` handle — a sampled batch's fresh
#     index keeps the atomics instead of paying a sort it never amortises.
# Building a plan reads nothing back to the host: `pygamd_index_guard` folds every out-of-range
# entry into a sentinel group behind the last real one (skipped, as the atomic kernels skip such
# rows) and raises the launch's error flag, which travels as PYGAMD_CHECK_INDEX says (async: flag
# ring; sync: one blocking read; off: nothing).  The hub split plan (one host read) is added the
# first time a cached plan is REUSED.  Plans are int32 whenever the sizes fit and the cache is
# bounded in BYTES (PYGAMD_SCATTER_PLAN_BYTES, default 1 GiB; a plan at the products shape is
# 0.26 GB), oldest out first; an entry also dies with its index tensor.
SORTED_SCATTER_MIN_ROWS = 1 << 16
SORTED_SCATTER_MIN_ELEMS = 1 << 23
SORTED_SCATTER_ALWAYS_ROWS = 1 << 20
SCATTER_PLAN_CACHE_BYTES = int(os.environ.get('PYGAMD_SCATTER_PLAN_BYTES', str(1 << 30)))
SCATTER_SEEN_ENTRIES = 64
_scatter_plans = collections.OrderedDict()  # key -> [weakref(base), version, plan, nbytes, hub?, oob]
_scatter_seen = collections.OrderedDict()   # key -> (weakref(base), version): met once, no plan
_INT32_MAX = (1 << 31) - 1


def _index_key(index: Tensor, dim_size: int):
    """Identity of an index tensor: the tensor it views (``edge_index[1]`` is a new view object on
    every call) + where in it + its version counter."""
    base = index._base if index._base is not None else index
    return base, (id(base), index.storage_offset(), index.numel(), index.stride(0), index.dtype,
                  int(dim_size))


def _build_scatter_plan(index: Tensor, dim_size: int, oob: list):
    """``oob``: a one-element list the flag ring sets to True when the guard's flag arrives
    raised — a CACHED plan then re-reports its out-of-range entries on every later hit (the
    kernels skip such rows either way; without the marker only the first use would say so)."""
    n = index.numel()
    small = n < _INT32_MAX and dim_size + 1 < _INT32_MAX
    ring, slot, err = _native._index_flag(index.device, True)
    keys = _native.index_guard(index, dim_size, err,
                               dtype=torch.int32 if small else torch.int64)

    def mark():
        oob[0] = True

    _native._index_flag_done(ring, slot, err, 'scatter', dim_size, index, on_flag=mark)
    sorted_keys, perm = _native.index_sort(keys, max_value=dim_size)
    # group `dim_size` = the out-of-range entries: it has a pointer entry and no output row
    ptr = _native.index2ptr(sorted_keys, dim_size + 1)[:dim_size + 1]
    if perm.dtype != ptr.dtype:
        perm = _native.cast_index(perm, ptr.dtype)
    return ptr, perm


def _plan_bytes(ptr: Tensor, perm: Tensor) -> int:
    return ptr.numel() * ptr.element_size() + perm.numel() * perm.element_size()


def _sorted_scatter_plan(index: Tensor, dim_size: int, force: bool = True):
    """(ptr [dim_size + 1], perm [n], hub) for ``index``, or None when the rule above leaves this
    call to the atomic kernels (``force=False`` and a first sighting)."""
    import weakref
    base, key = _index_key(index, dim_size)
    hit = _scatter_plans.get(key)
    if hit is not None and hit[0]() is base and hit[1] == index._version:
        _scatter_plans.move_to_end(key)
        if hit[5][0] and _native.INDEX_CHECK != 'off':  # known bad since an earlier use: say so
            _native._raise_out_of_range(index, dim_size, 'scatter',   # again, at the call
                                        _native._error_style.value)
        if not hit[4]:  # reused: now the hub split pays (one host read, once)
            ptr, perm, _ = hit[2]
            hit[2], hit[4] = (ptr, perm, _native.hub_plan(ptr)), True
        return hit[2]
    if hit is not None:
        _scatter_plans.pop(key, None)
    if not force and not isinstance(base, EdgeIndex):
        seen = _scatter_seen.get(key)
        if seen is None or seen[0]() is not base or seen[1] != index._version:
            _scatter_seen[key] = (weakref.ref(base, lambda _, k=key: _scatter_seen.pop(k, None)),
                                  index._version)
            while len(_scatter_seen) > SCATTER_SEEN_ENTRIES:
                _scatter_seen.popitem(last=False)
            return None
    _scatter_seen.pop(key, None)
    oob = [False]
    ptr, perm = _build_scatter_plan(index, dim_size, oob)
    plan = (ptr, perm, None)
    nbytes = _plan_bytes(ptr, perm)
    if nbytes <= SCATTER_PLAN_CACHE_BYTES:
        held = sum(e[3] for e in _scatter_plans.values())
        while _scatter_plans and held + nbytes > SCATTER_PLAN_CACHE_BYTES:
            held -= _scatter_plans.popitem(last=False)[1][3]
        _scatter_plans[key] = [weakref.ref(base, lambda _, k=key: _scatter_plans.pop(k, None)),
                               index._version, plan, nbytes, False, oob]
    return plan


def _use_sorted_scatter(rows: Tensor, index: Tensor, reduce: str) -> bool:
    return (reduce in ('sum', 'mean', 'min', 'max') and index.numel() >= SORTED_SCATTER_MIN_ROWS
            and rows.numel() >= SORTED_SCATTER_MIN_ELEMS and rows.dtype == torch.float32
            and not torch.cuda.is_current_stream_capturing())


def _scatter_rows_auto(rows: Tensor, index: Tensor, dim_size: int, reduce: str,
                       return_count: bool = False):
    """`_native.scatter_rows` semantics; large inputs through the sorted route."""
    plan = None
    if _use_sorted_scatter(rows, index, reduce):
        plan = _sorted_scatter_plan(index, dim_size,
                                    force=index.numel() >= SORTED_SCATTER_ALWAYS_ROWS)
    if plan is None:
        return _native.scatter_rows(rows, index, dim_size, reduce, return_count=return_count)
    ptr, perm, hub = plan
    out = _native.spmm_csr(ptr, perm, rows, reduce, n_rows=dim_size, hub=hub)
    if return_count:
        return out, (ptr[1:] - ptr[:-1]).to(torch.float32)
    return out


class GatherFunction(Function):
    """``x.index_select(0, index)`` (message_passing.py:263-290); backward = scatter-add."""

    @staticmethod
    def forward(ctx, x: Tensor, index: Tensor, check_bounds: bool):
        ctx.save_for_backward(index)
        ctx.x_shape = x.shape
        out = _native.gather_rows(_rows(x), index, check_bounds)
        return _shaped(out, (index.numel(), *x.shape[1:]))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        (index,) = ctx.saved_tensors
        g = _scatter_rows_auto(_rows(grad_out), index, ctx.x_shape[0], 'sum')
        return g.view(ctx.x_shape), None, None


class ScatterFunction(Function):
    """``scatter(src, index, 0, dim_size, reduce)`` on an unsorted index (utils/_scatter.py)."""

    @staticmethod
    def forward(ctx, src: Tensor, index: Tensor, dim_size: int, reduce: str):
        s2 = _rows(src)
        ctx.reduce, ctx.src_shape = reduce, src.shape
        if reduce == 'mean':
            out, count = _scatter_rows_auto(s2, index, dim_size, reduce, return_count=True)
            ctx.save_for_backward(index, count)
        elif reduce in ('min', 'max', 'mul'):
            # the RETURNED tensor is what the backward compares against: saving it (not its 2-D
            # alias) lets autograd detect an in-place edit of the result
            res = _shaped(_scatter_rows_auto(s2, index, dim_size, reduce),
                          (dim_size, *src.shape[1:]))
            ctx.save_for_backward(index, s2, res)
            return res
        else:
            out = _scatter_rows_auto(s2, index, dim_size, reduce)
            ctx.save_for_backward(index)
        return _shaped(out, (dim_size, *src.shape[1:]))

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        reduce = ctx.reduce
        # the forward's index flag has normally arrived by now: raise before gathering with an
        # index the forward found out of range (no wait: INDEX_CHECK='async')
        _native.poll_index_errors()
        g2 = _rows(grad_out)
        if reduce in ('sum', 'add'):
            (index,) = ctx.saved_tensors
            grad = _native.gather_rows(g2, index)
        elif reduce == 'mean':
            index, count = ctx.saved_tensors
            grad = _native.gather_rows(g2 / count.clamp(min=1).view(-1, 1), index)
        elif reduce in ('min', 'max'):
            index, s2, out = ctx.saved_tensors
            grad = _native.scatter_minmax_backward(s2, index, _rows(out), g2)
        elif reduce == 'mul':
            # ATen's scatter_reduce 'prod' backward incl. its zero-count rule (one zero in the
            # group: that element gets g * prod(others); two or more: all 0) — g * out / src alone
            # would be NaN there
            index, s2, out = ctx.saved_tensors
            grad = _native.scatter_mul_backward(s2, index, _rows(out), g2)
        else:
            raise NotImplementedError(f"backward of scatter(reduce='{reduce}') is undefined")
        return grad.view(ctx.src_shape), None, None, None


class MultiReduceFunction(Function):
    """Several reductions of the same grouped rows in ONE read (``pygamd_multi_reduce_csr``; the
    shared work of nn/aggr/fused.py:191-336): returns one ``[dim_size, F]`` tensor per name in
    ``want`` (names from ``sum | pow_sum | min | max``).  ``perm`` is the stable sort permutation
    of ``index`` (``None`` when the rows are already grouped).  Backward composes the existing
    kernels: a gather for ``sum``, ``2 x`` times a gather for ``pow_sum`` (skipped with
    ``semi_grad``), the tie-splitting min / max gradient of the reference."""

    @staticmethod
    def forward(ctx, x: Tensor, index: Tensor, ptr: Tensor, perm: Optional[Tensor], want: tuple,
                semi_grad: bool):
        outs = _native.multi_reduce_csr(ptr, perm, x, want)
        ctx.want, ctx.semi_grad = want, semi_grad
        ctx.save_for_backward(x, index, *[outs[k] for k in want if k in ('min', 'max')])
        return tuple(outs[k] for k in want)

    @staticmethod
    def backward(ctx, *grads):
        x, index, *extrema = ctx.saved_tensors
        extrema = iter(extrema)
        total = None
        for name, g in zip(ctx.want, grads):
            out = next(extrema) if name in ('min', 'max') else None
            if g is None:
                continue
            g = g.contiguous()
            if name == 'sum':
                term = _native.gather_rows(g, index)
            elif name == 'pow_sum':
                if ctx.semi_grad:
                    continue
                term = _native.gather_rows(g, index).mul_(x).mul_(2.0)
            else:
                term = _native.scatter_minmax_backward(x, index, out, g)
            total = term if total is None else total.add_(term)
        return total, None, None, None, None, None


class SegmentFunction(Function):
    """``segment(src, ptr, reduce)`` over contiguous row ranges (utils/_segment.py:11-50)."""

    @staticmethod
    def forward(ctx, src: Tensor, ptr: Tensor, reduce: str):
        s2 = _rows(src)
        n_seg = ptr.numel() - 1
        out = _native.spmm_csr(ptr, None, s2, reduce, n_rows=n_seg)
        ctx.reduce, ctx.src_shape = reduce, src.shape
        res = _shaped(out, (n_seg, *src.shape[1:]))
        if reduce in ('min', 'max'):
            ctx.save_for_backward(ptr, s2, res)
        else:
            ctx.save_for_backward(ptr)
        return res

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        reduce = ctx.reduce
        g2 = _rows(grad_out)
        n = ctx.src_shape[0]
        if reduce in ('min', 'max'):
            ptr, s2, out = ctx.saved_tensors
            out = _rows(out)
            ntie = _native.spmm_tie_count(ptr, None, s2, out, count_self=False)
            index = _native.ptr2index(ptr, n)
            # each row belongs to exactly one segment: grad = [src == out[seg]] * g[seg] / ntie[seg]
            o_e = _native.gather_rows(out, index)
            # ATen's _segment_reduce backward (the reference's CPU path, utils/_segment.py:48)
            # averages over the tied extrema ONLY where the incoming gradient is positive
            # (SegmentReduce.cpp: `if (grad_input > 0) grad_input /= counter`); a negative
            # gradient reaches every tied element undivided.  Matched as is.
            g_e = _native.gather_rows(torch.where(g2 > 0, g2 / ntie.clamp(min=1), g2), index)
            grad = torch.where(s2 == o_e, g_e, torch.zeros_like(g_e))
        else:
            (ptr,) = ctx.saved_tensors
            index = _native.ptr2index(ptr, n)
            if reduce == 'mean':
                cnt = (ptr[1:] - ptr[:-1]).clamp(min=1).to(torch.float32)
                g2 = g2 / cnt.view(-1, 1)
            grad = _native.gather_rows(g2, index)
        return grad.view(ctx.src_shape), None, None


class SegmentSoftmaxFunction(Function):
    """Softmax within ``ptr`` segments (utils/_softmax.py:60-81); the max is taken on the detached
    input, so the backward is the plain softmax Jacobian."""

    @staticmethod
    def forward(ctx, src: Tensor, ptr: Tensor):
        s2 = _rows(src)
        res = _shaped(_native.segment_softmax_forward(s2, ptr), src.shape)
        ctx.save_for_backward(res, ptr)
        ctx.src_shape = src.shape
        return res

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        out, ptr = ctx.saved_tensors
        out = _rows(out)
        g = _native.segment_softmax_backward(out, grad_out.reshape(out.shape), ptr)
        return g.view(ctx.src_shape), None


class IndexSoftmaxFunction(Function):
    """Softmax within the groups of an UNSORTED ``index`` (utils/_softmax.py:82-88: the maximum is
    taken on the detached input, ``1e-16`` joins the denominator) as dedicated kernels instead of
    the six-pass scatter / gather composition; backward = the softmax Jacobian,
    ``out * (g - sum_group(out * g))``."""

    @staticmethod
    def forward(ctx, src: Tensor, index: Tensor, num_groups: int):
        res = _shaped(_native.softmax_index_forward(_rows(src), index, num_groups), src.shape)
        ctx.save_for_backward(res, index)
        ctx.num_groups = num_groups
        return res

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        out, index = ctx.saved_tensors
        o2 = _rows(out)
        g = _native.softmax_index_backward(o2, grad_out.reshape(o2.shape), index, ctx.num_groups)
        return g.view(out.shape), None, None


class SegmentLogSumExpFunction(Function):
    """``segment_logsumexp(src, ptr, dim=0)`` (utils/_segment.py:53-80) for 2-D ``src``; the total
    derivative through the subtracted maximum is the in-segment softmax."""

    @staticmethod
    def forward(ctx, src: Tensor, ptr: Tensor):
        s2 = _rows(src)
        res = _shaped(_native.segment_logsumexp_forward(s2, ptr),
                      (ptr.numel() - 1, *src.shape[1:]))
        ctx.save_for_backward(s2, res, ptr)
        ctx.src_shape = src.shape
        return res

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        s2, out, ptr = ctx.saved_tensors
        g = _native.segment_logsumexp_backward(s2, _rows(out), _rows(grad_out), ptr)
        return g.view(ctx.src_shape), None


class GatEdgeSoftmaxFunction(Function):
    """alpha[k,h] = softmax_row(leaky_relu(alpha_src[col[k],h] + alpha_dst[i,h])) in by-destination
    slot order (nn/conv/gat_conv.py:387-406)."""

    @staticmethod
    def forward(ctx, alpha_src: Tensor, alpha_dst: Tensor, graph: EdgeIndex, slope: float):
        fwd = graph.by_dst()
        alpha = _native.gat_edge_softmax_forward(fwd.ptr, fwd.idx, alpha_src, alpha_dst, slope)
        ctx.save_for_backward(alpha_src, alpha_dst, alpha)
        ctx.graph, ctx.slope = graph, slope
        return alpha

    @staticmethod
    def backward(ctx, grad_alpha: Tensor):
        alpha_src, alpha_dst, alpha = ctx.saved_tensors
        fwd = ctx.graph.by_dst()
        g_src, g_dst = _native.gat_edge_softmax_backward(fwd.ptr, fwd.idx, alpha_src, alpha_dst,
                                                         alpha, grad_alpha, ctx.slope)
        return g_src, g_dst, None, None


# PYGAMD_GAT_FUSED_BWD=0: the SDDMM and the transposed SpMM of the GAT backward as two launches
GAT_FUSED_BACKWARD = os.environ.get('PYGAMD_GAT_FUSED_BWD', '1') != '0'


class GatAttendFunction(Function):
    """One GAT attention + aggregation step on projected features ``x [N, H, C]`` (source and
    destination features are the same tensor): node terms ``(x * att).sum(-1)``, edge logits +
    leaky ReLU + softmax per destination, weighted aggregation (gat_conv.py:330-332, 387-408) —
    the three kernels of HeadDotFunction / GatEdgeSoftmaxFunction / SpmmFunction under ONE
    autograd node, so that the two gradients of ``x`` (through the aggregation and through the
    node terms) meet inside ``head_dot_bwd_kernel`` instead of in a separate add pass."""

    @staticmethod
    def forward(ctx, x: Tensor, att_src: Tensor, att_dst: Tensor, graph: EdgeIndex,
                slope: float, n_dst: int):
        N, H, C = x.shape
        x2 = x.reshape(N, H * C)
        a_src, a_dst = _native.head_dot_forward(x2, att_src.reshape(-1), att_dst.reshape(-1),
                                                H, C)
        fwd = graph.by_dst()
        a_dst = a_dst[:n_dst].contiguous()
        alpha = _native.gat_edge_softmax_forward(fwd.ptr, fwd.idx, a_src, a_dst, slope)
        out = _native.spmm_csr(fwd.ptr, fwd.idx, x2, 'sum', n_rows=fwd.n_rows, w=alpha,
                               hub=fwd.hub)
        ctx.save_for_backward(x2, att_src, att_dst, a_src, a_dst, alpha)
        ctx.graph, ctx.slope, ctx.dims = graph, slope, (N, H, C)
        return out.view(fwd.n_rows, H, C)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: Tensor):
        x2, att_src, att_dst, a_src, a_dst, alpha = ctx.saved_tensors
        N, H, C = ctx.dims
        graph = ctx.graph
        fwd, bwd = graph.by_dst(), graph.by_src()
        g2 = _rows(grad_out)
        if GAT_FUSED_BACKWARD:
            # ONE pass over the by-source slots gathers every gradient row once for both
            #   d alpha[k, h] = <grad_out[i, h, :], x[j, h, :]>  (filed under the edge's
            #   destination slot, where the softmax backward reads it) and
            #   d x through the aggregation = sum_i alpha * grad_out[i]
            # (two launches — SDDMM by destination + transposed weighted SpMM — gathered 2 x
            # E rows of H * C floats: 1.39 of config 3's 5.5 ms per step)
            grad_alpha, grad_x = _native.sddmm_spmm_csr(
                bwd.ptr, bwd.idx, graph.src_slot_to_dst_slot(), x2, g2, alpha, fwd.nnz, H)
        else:
            # d x through the aggregation: the same coefficients on the transposed handle
            grad_x = _native.spmm_csr(bwd.ptr, bwd.idx, g2, 'sum', n_rows=bwd.n_rows,
                                      eid=graph.src_slot_to_dst_slot(), w=alpha, hub=bwd.hub)
            # d alpha[k, h] = <grad_out[i, h, :], x[j, h, :]>, then back through the edge softmax
            grad_alpha = _native.sddmm_csr(fwd.ptr, fwd.idx, None, g2, x2, fwd.nnz, H)
        g_src, g_dst = _native.gat_edge_softmax_backward(fwd.ptr, fwd.idx, a_src, a_dst, alpha,
                                                         grad_alpha.view(alpha.shape), ctx.slope)
        if g_dst.size(0) < N:  # destinations are a prefix of the nodes
            pad = g_dst.new_zeros(N, H)
            pad[:g_dst.size(0)] = g_dst
            g_dst = pad
        # d x through the node terms is ADDED to grad_x by the kernel that also reduces d att
        _, g_att_src, g_att_dst = _native.head_dot_backward(
            x2, att_src.reshape(-1), att_dst.reshape(-1), g_src, g_dst, H, C, True,
            accumulate_into=grad_x)
        return (grad_x.view(N, H, C), g_att_src.view(att_src.shape),
                g_att_dst.view(att_dst.shape), None, None, None)


class HeadDotFunction(Function):
    """(a_src, a_dst) = ((x * att_src).sum(-1), (x * att_dst).sum(-1)) for x [N, H, C] and
    att_* [1, H, C] (nn/conv/gat_conv.py:330-332) — one pass over x, one fused backward."""

    @staticmethod
    def forward(ctx, x: Tensor, att_a: Tensor, att_b: Optional[Tensor]):
        N, H, C = x.shape
        out_a, out_b = _native.head_dot_forward(x.reshape(N, H * C), att_a.reshape(-1),
                                                None if att_b is None else att_b.reshape(-1),
                                                H, C)
        if att_b is None:
            raise ValueError('HeadDotFunction needs both attention vectors')
        ctx.save_for_backward(x, att_a, att_b)
        ctx.has_b = True
        return out_a, out_b

    @staticmethod
    def backward(ctx, grad_a: Tensor, grad_b: Tensor):
        x, att_a, att_b = ctx.saved_tensors
        N, H, C = x.shape
        if grad_a is None:
            grad_a = x.new_zeros(N, H)
        if ctx.has_b and grad_b is None:
            grad_b = x.new_zeros(N, H)
        gx, ga, gb = _native.head_dot_backward(
            x.reshape(N, H * C), att_a.reshape(-1), None if att_b is None else att_b.reshape(-1),
            grad_a, grad_b if ctx.has_b else None, H, C, ctx.needs_input_grad[0])
        return (None if gx is None else gx.view(N, H, C), ga.view(att_a.shape),
                None if gb is None else gb.view(att_b.shape))


class LinearFunction(Function):
    """``x @ weight.T + bias`` (nn/dense/linear.py:121-127 ``F.linear``) on the fp32-MFMA kernels
    of csrc/gemm.hip: forward with the bias epilogue, input gradient with the same kernel on the
    transposed weight, weight gradient as a deterministic split reduction that also returns the
    bias gradient (column sums of ``grad_out``) from the same pass."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor]):
        x2 = x.reshape(-1, x.size(-1))
        ctx.save_for_backward(x2, weight)
        ctx.x_shape, ctx.has_bias = x.shape, bias is not None
        out = _native.linear_forward(x2, weight, bias)
        return _shaped(out, (*x.shape[:-1], weight.size(0)))

    @staticmethod
    @once_differentiable  # the kernels record no graph: create_graph=True raises instead of
    def backward(ctx, grad_out: Tensor):  # silently dropping the second derivative
        x2, weight = ctx.saved_tensors
        g2 = grad_out.reshape(-1, grad_out.size(-1))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _native.linear_dgrad(g2, weight.t().contiguous()).view(ctx.x_shape)
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw = _native.linear_wgrad(g2, x2, bias_grad=need_b)  # column sums from the same pass
            if need_b:
                gw, gb = gw
        elif need_b:
            gb = _native.colsum(g2)
        return gx, gw, gb


# Below this many rows the library GEMM stays (F.linear): a Python-side autograd Function per call
# costs more than such a product.  From here up the own kernels fill the chip: 128-row tiles for
# the full-batch layers, 64 x 64 tiles + a deterministic split over the reduction for sampled
# blocks and Cora-sized inputs (csrc/gemm.hip nt_shape; round 3 drew this line at 16,384 rows).
OWN_GEMM_MIN_ROWS = 1024


def own_linear_eligible(x: Tensor, weight: Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.dim() >= 2 and x.numel() // max(x.size(-1), 1) >= OWN_GEMM_MIN_ROWS
            and not torch.jit.is_scripting() and not torch.is_autocast_enabled())


# The autograd nodes of csrc/torch_binding.cpp (LinearAG / SpmmAG / BiasActAG): the same kernels in
# the same order as LinearFunction / SpmmFunction / BiasActFunction below, with forward AND backward
# running in C++ — an eager, launch-bound step (2-layer GCN at the Cora shape) spends most of its
# host time in the Python Functions and their marshalling (scripts/eager_probe.py).  Taken for
# plain float32 operands when nothing beyond those nodes is asked for; PYGAMD_CPP_AUTOGRAD=0, a
# missing binding, torch.compile tracing or bench.py's per-launch timing sink keep the Python nodes.
CPP_AUTOGRAD = os.environ.get('PYGAMD_CPP_AUTOGRAD', '1') != '0'


def _cpp_nodes():
    if not CPP_AUTOGRAD or _native.timing_sink is not None or torch.compiler.is_compiling():
        return None
    from . import _compiled
    return _compiled.ops()


def _plain_f32(*ts) -> bool:
    return all(t is None or (type(t) in (Tensor, torch.nn.Parameter) and t.is_cuda
                             and t.dtype == torch.float32) for t in ts)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """``F.linear`` semantics (``weight [out, in]``); large float32 HIP inputs run on csrc/gemm.hip."""
    if own_linear_eligible(x, weight):
        C = _cpp_nodes()
        if C is not None and _plain_f32(x, weight, bias) and x.size(-1) == weight.size(1) \
                and (bias is None or bias.numel() == weight.size(0)):
            return C.linear_ag(x, weight, bias)
        return LinearFunction.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def spmm_node(x: Tensor, w: Optional[Tensor], graph: EdgeIndex, reduce: str,
              w_order: str) -> Tensor:
    """``SpmmFunction.apply`` semantics; sum / mean with no gradient asked for the weights run as
    the C++ autograd node."""
    C = _cpp_nodes()
    # The C++ node takes BOTH sorted forms up front.  The by-source one (a transposed sort + a hub
    # plan with a host read, per fresh graph) only serves the gradient of `x`: inference under
    # no_grad and first layers whose `x` takes no gradient keep the Python node, which builds it
    # inside backward() when needs_input_grad[0] says so.
    if (C is not None and reduce in ('sum', 'mean') and _plain_f32(x, w) and x.dim() >= 2
            and torch.is_grad_enabled() and x.requires_grad
            and x.numel() > 0 and x.size(0) == graph.num_src_nodes and not graph.atomic_backward
            and (w is None or (w.dim() == 1 and not w.requires_grad
                               and w.numel() == graph.num_edges))):
        fwd, bwd = graph.by_dst(), graph.by_src()
        eid = eid_t = None
        if w is not None:
            eid = fwd.perm if w_order == 'coo' else None
            eid_t = bwd.perm if w_order == 'coo' else graph.src_slot_to_dst_slot()
        (h_rows, h_cptr, n_hub, n_chunks), (t_rows, t_cptr, t_hub, t_chunks) = fwd.hub, bwd.hub
        inv = fwd.inv_degree() if reduce == 'mean' else None
        return C.spmm_ag(x, w, fwd.ptr, fwd.idx, eid, h_rows, h_cptr, n_hub, n_chunks, bwd.ptr,
                         bwd.idx, eid_t, t_rows, t_cptr, t_hub, t_chunks, inv,
                         _native.REDUCE_IDS[reduce], _native.HUB_THRESHOLD, _native.HUB_CHUNK)
    return SpmmFunction.apply(x, w, graph, reduce, w_order)


class BiasActFunction(Function):
    """``relu(x + bias)`` (or ``x + bias``) in one pass, backward in one pass: the masked gradient
    and — from the same read — its column sums, the bias gradient (``pygamd_bias_act`` /
    ``pygamd_relu_backward_colsum``).  Replaces ``out + self.bias`` of a conv layer
    (gat_conv.py:378-385, gcn_conv.py:278-281) followed by the model's ReLU
    (basic_gnn.py:262-263): four ATen passes forward + backward become two."""

    @staticmethod
    def forward(ctx, x: Tensor, bias: Optional[Tensor], relu: bool):
        x2 = x.reshape(-1, x.size(-1))
        out = _native.bias_act(x2, bias, relu)
        ctx.relu, ctx.has_bias, ctx.shape = relu, bias is not None, x.shape
        res = _shaped(out, x.shape)
        if relu:
            ctx.save_for_backward(res)
        return res

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: Tensor):
        g2 = grad_out.reshape(-1, grad_out.size(-1))
        need_b = ctx.has_bias and ctx.needs_input_grad[1]
        if ctx.relu:
            (out, ) = ctx.saved_tensors
            g, gb = _native.relu_backward_colsum(g2, out.reshape(-1, out.size(-1)), need_b)
            return g.view(ctx.shape), (gb if need_b else None), None
        return grad_out, (_native.colsum(g2) if need_b else None), None


def bias_act(x: Tensor, bias: Optional[Tensor], relu: bool) -> Tensor:
    """``relu(x + bias)`` / ``x + bias``; float32 device tensors take the one-pass kernels."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.numel() > 0 \
            and (bias is None or (bias.dtype == torch.float32 and bias.dim() == 1)) \
            and not torch.jit.is_scripting() and not torch.is_autocast_enabled():
        C = _cpp_nodes()
        if C is not None and _plain_f32(x, bias) and (bias is None or bias.numel() == x.size(-1)):
            return C.bias_act_ag(x, bias, bool(relu))
        return BiasActFunction.apply(x, bias, relu)
    out = x if bias is None else x + bias
    return out.relu() if relu else out


class CrossEntropyRowsFunction(Function):
    """``F.cross_entropy(logits[index], target[index])`` — the loss a full-batch model takes on its
    training split (``examples/ogbn_train.py``-style ``out[train_idx]``) — as ONE pass over the
    selected rows forward (no gathered copy of the rows, no separate log-softmax / NLL launches)
    and a row scatter backward: seven ATen launches (0.75 ms at the ogbn-products shape, whose
    ``nll_loss`` reductions run on one workgroup) become two + three."""

    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, index: Optional[Tensor]):
        loss, grad_rows = _native.cross_entropy_rows(logits, target, index)
        ctx.save_for_backward(grad_rows, index)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        grad_rows, index = ctx.saved_tensors
        scaled = grad_rows * g
        if index is None:
            return scaled, None, None
        dense = torch.zeros(ctx.shape, dtype=torch.float32, device=scaled.device)
        dense.index_add_(0, index, scaled)   # (duplicate indices add up, as out[index] would)
        return dense, None, None


def cross_entropy(logits: Tensor, target: Tensor, index: Optional[Tensor] = None) -> Tensor:
    """Mean cross entropy of ``logits[index]`` against ``target[index]`` (``index`` None: every
    row); ``target``: one int64 label per row of ``logits``.  Float32 HIP logits take the one-pass
    kernel; anything else (CPU tensors, other dtypes, class weights are not offered) runs
    ``F.cross_entropy`` on the gathered rows."""
    if (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2
            and logits.size(0) > 0 and logits.size(1) > 0 and logits.stride(1) == 1
            and target.dtype == torch.int64 and target.dim() == 1
            and target.numel() == logits.size(0)
            and (index is None or (index.dtype == torch.int64 and index.dim() == 1
                                   and index.numel() > 0))
            and not torch.cuda.is_current_stream_capturing()):
        return CrossEntropyRowsFunction.apply(
            logits, target.contiguous(), None if index is None else index.contiguous())
    if index is None:
        return torch.nn.functional.cross_entropy(logits, target)
    return torch.nn.functional.cross_entropy(logits[index], target[index])

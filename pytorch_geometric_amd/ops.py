"""``torch.ops.pyg_amd.*`` — the kernels as registered PyTorch operators (seam S2 of SURVEY.md §8(b)).

The reference reaches its native code through operator SCHEMAS it expects to exist:
``torch.ops.torch_sparse.spmm_{sum,mean,min,max}`` (edge_index.py:1798-1810),
``torch_scatter.segment_csr`` / ``scatter`` (utils/_segment.py:34, utils/_scatter.py:104),
``pyg_lib.ops.softmax_csr`` (utils/_softmax.py:58), ``pyg_lib.ops.index_sort``
(utils/_index_sort.py:32).  This module registers the MI355X counterparts under the ``pyg_amd``
namespace with ``torch.library``:

* a device implementation (HIP, through the C ABI) for ``cuda`` tensors only — there is no CPU
  kernel, a CPU tensor fails in the dispatcher;
* a FAKE (meta) kernel per operator, so ``FakeTensorMode`` / ``torch.compile`` / ``torch.export``
  can propagate shapes and dtypes without running anything;
* autograd through ``torch.library.register_autograd``: every differentiable operator is paired
  with an opaque ``*_backward`` operator (itself registered with a fake kernel), so AOTAutograd can
  trace forward and backward graphs — under ``torch.compile`` the kernels stay single opaque nodes
  instead of forcing the backend to step aside (round-1 behaviour).

Operators (all index tensors int32 / int64, features float32):

==========================  ===========================================================
``index_sort``              ``(Tensor inputs, int? max_value) -> (Tensor, Tensor)``
``index2ptr`` / ``ptr2index``  ``(Tensor, int) -> Tensor``
``gather``                  ``(Tensor x, Tensor index) -> Tensor``           (index_select dim 0)
``scatter``                 ``(Tensor src, Tensor index, int dim_size, str reduce) -> Tensor``
``segment_csr``             ``(Tensor src, Tensor ptr, str reduce) -> Tensor``
``softmax_csr``             ``(Tensor src, Tensor ptr) -> Tensor``
``spmm``                    ``(Tensor rowptr, Tensor col, Tensor? value, Tensor other, str reduce)``
``linear``                  ``(Tensor x, Tensor weight, Tensor? bias) -> Tensor``
==========================  ===========================================================
"""
import math
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op, register_autograd

from . import _native

_DEV = 'cuda'
_REDUCES = ('sum', 'mean', 'min', 'max', 'mul')


def _rows(t: Tensor) -> Tensor:
    """[n, ...] -> [n, prod(...)] (well defined for empty tensors too)."""
    return t.reshape(t.size(0), math.prod(t.shape[1:]))


# ---- integer side (no gradients) ---------------------------------------------------------------
@custom_op('pyg_amd::index_sort', mutates_args=(), device_types=_DEV)
def index_sort(inputs: Tensor, max_value: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    return _native.index_sort(inputs, max_value)


@index_sort.register_fake
def _(inputs, max_value=None):
    return torch.empty_like(inputs), torch.empty(inputs.shape, dtype=torch.int64,
                                                 device=inputs.device)


@custom_op('pyg_amd::index2ptr', mutates_args=(), device_types=_DEV)
def index2ptr(index: Tensor, size: int) -> Tensor:
    return _native.index2ptr(index, size)


@index2ptr.register_fake
def _(index, size):
    return index.new_empty(size + 1)


@custom_op('pyg_amd::ptr2index', mutates_args=(), device_types=_DEV)
def ptr2index(ptr: Tensor, n: int) -> Tensor:
    return _native.ptr2index(ptr, n)


@ptr2index.register_fake
def _(ptr, n):
    return ptr.new_empty(n)


# ---- gather / scatter ----------------------------------------------------------------------------
@custom_op('pyg_amd::gather', mutates_args=(), device_types=_DEV)
def gather(x: Tensor, index: Tensor) -> Tensor:
    out = _native.gather_rows(_rows(x), index)
    return out.reshape(index.numel(), *x.shape[1:])


@gather.register_fake
def _(x, index):
    return x.new_empty(index.numel(), *x.shape[1:])


@custom_op('pyg_amd::scatter', mutates_args=(), device_types=_DEV)
def scatter(src: Tensor, index: Tensor, dim_size: int, reduce: str) -> Tensor:
    if reduce not in _REDUCES:
        raise ValueError(f"Encountered invalid `reduce` argument '{reduce}'")
    out = _native.scatter_rows(_rows(src), index, dim_size, reduce)
    return out.reshape(dim_size, *src.shape[1:])


@scatter.register_fake
def _(src, index, dim_size, reduce):
    return src.new_empty(dim_size, *src.shape[1:])


@custom_op('pyg_amd::scatter_backward', mutates_args=(), device_types=_DEV)
def scatter_backward(grad: Tensor, src: Tensor, index: Tensor, out: Tensor,
                     reduce: str) -> Tensor:
    g2 = _rows(grad).contiguous()
    s2 = _rows(src)
    if reduce == 'sum':
        res = _native.gather_rows(g2, index)
    elif reduce == 'mean':
        ones = torch.ones(index.numel(), 1, dtype=torch.float32, device=src.device)
        cnt = _native.scatter_rows(ones, index, grad.size(0), 'sum').clamp_(min=1)
        res = _native.gather_rows(g2 / cnt, index)
    elif reduce in ('min', 'max'):
        res = _native.scatter_minmax_backward(s2, index, _rows(out), g2)
    else:
        res = _native.scatter_mul_backward(s2, index, _rows(out), g2)
    return res.reshape(src.shape)


@scatter_backward.register_fake
def _(grad, src, index, out, reduce):
    return torch.empty_like(src)


def _scatter_setup(ctx, inputs, output):
    src, index, dim_size, reduce = inputs
    ctx.reduce = reduce
    ctx.save_for_backward(src, index, output)


def _scatter_bwd(ctx, grad):
    src, index, out = ctx.saved_tensors
    return scatter_backward(grad, src, index, out, ctx.reduce), None, None, None


register_autograd('pyg_amd::scatter', _scatter_bwd, setup_context=_scatter_setup)


def _gather_setup(ctx, inputs, output):
    x, index = inputs
    ctx.n = x.size(0)
    ctx.save_for_backward(index)


def _gather_bwd(ctx, grad):
    (index, ) = ctx.saved_tensors
    return scatter(grad.contiguous(), index, ctx.n, 'sum'), None


register_autograd('pyg_amd::gather', _gather_bwd, setup_context=_gather_setup)


# ---- segment_csr / softmax_csr -------------------------------------------------------------------
@custom_op('pyg_amd::segment_csr', mutates_args=(), device_types=_DEV)
def segment_csr(src: Tensor, ptr: Tensor, reduce: str) -> Tensor:
    if reduce not in ('sum', 'mean', 'min', 'max'):
        raise ValueError(f"Encountered invalid `reduce` argument '{reduce}'")
    n_seg = ptr.numel() - 1
    out = _native.spmm_csr(ptr, None, _rows(src), reduce, n_rows=n_seg)
    return out.reshape(n_seg, *src.shape[1:])


@segment_csr.register_fake
def _(src, ptr, reduce):
    return src.new_empty(ptr.numel() - 1, *src.shape[1:])


@custom_op('pyg_amd::segment_csr_backward', mutates_args=(), device_types=_DEV)
def segment_csr_backward(grad: Tensor, src: Tensor, ptr: Tensor, out: Tensor,
                         reduce: str) -> Tensor:
    n = src.size(0)
    g2 = _rows(grad).contiguous()
    index = _native.ptr2index(ptr, n)
    if reduce in ('min', 'max'):
        s2, o2 = src.reshape(n, -1), _rows(out)
        ntie = _native.spmm_tie_count(ptr, None, s2, o2, count_self=False)
        o_e = _native.gather_rows(o2, index)
        # (ATen averages over ties only where the gradient is positive — see SegmentFunction)
        g_e = _native.gather_rows(torch.where(g2 > 0, g2 / ntie.clamp(min=1), g2), index)
        res = torch.where(s2 == o_e, g_e, torch.zeros_like(g_e))
    else:
        if reduce == 'mean':
            g2 = g2 / (ptr[1:] - ptr[:-1]).clamp(min=1).to(torch.float32).view(-1, 1)
        res = _native.gather_rows(g2, index)
    return res.reshape(src.shape)


@segment_csr_backward.register_fake
def _(grad, src, ptr, out, reduce):
    return torch.empty_like(src)


def _segment_setup(ctx, inputs, output):
    src, ptr, reduce = inputs
    ctx.reduce = reduce
    ctx.save_for_backward(src, ptr, output)


def _segment_bwd(ctx, grad):
    src, ptr, out = ctx.saved_tensors
    return segment_csr_backward(grad, src, ptr, out, ctx.reduce), None, None


register_autograd('pyg_amd::segment_csr', _segment_bwd, setup_context=_segment_setup)


@custom_op('pyg_amd::softmax_csr', mutates_args=(), device_types=_DEV)
def softmax_csr(src: Tensor, ptr: Tensor) -> Tensor:
    out = _native.segment_softmax_forward(_rows(src), ptr)
    return out.reshape(src.shape)


@softmax_csr.register_fake
def _(src, ptr):
    return torch.empty_like(src)


@custom_op('pyg_amd::softmax_csr_backward', mutates_args=(), device_types=_DEV)
def softmax_csr_backward(out: Tensor, grad: Tensor, ptr: Tensor) -> Tensor:
    o2 = _rows(out)
    return _native.segment_softmax_backward(o2, grad.reshape(o2.shape), ptr).reshape(out.shape)


@softmax_csr_backward.register_fake
def _(out, grad, ptr):
    return torch.empty_like(out)


def _softmax_setup(ctx, inputs, output):
    ctx.save_for_backward(output, inputs[1])


def _softmax_bwd(ctx, grad):
    out, ptr = ctx.saved_tensors
    return softmax_csr_backward(out, grad.contiguous(), ptr), None


register_autograd('pyg_amd::softmax_csr', _softmax_bwd, setup_context=_softmax_setup)


# ---- spmm on a CSR pair (rows = destinations) ---------------------------------------------------------
@custom_op('pyg_amd::spmm', mutates_args=(), device_types=_DEV)
def spmm(rowptr: Tensor, col: Tensor, value: Optional[Tensor], other: Tensor,
         reduce: str) -> Tensor:
    if reduce not in ('sum', 'mean', 'min', 'max'):
        raise ValueError(f"`reduce` argument '{reduce}' not supported")
    if value is not None and reduce in ('min', 'max'):
        raise NotImplementedError('edge weights are not supported for min/max')
    n_rows = rowptr.numel() - 1
    out = _native.spmm_csr(rowptr, col, _rows(other), reduce, n_rows=n_rows,
                           w=value, hub=_native.hub_plan(rowptr))
    return out.reshape(n_rows, *other.shape[1:])


@spmm.register_fake
def _(rowptr, col, value, other, reduce):
    return other.new_empty(rowptr.numel() - 1, *other.shape[1:])


@custom_op('pyg_amd::spmm_backward', mutates_args=(), device_types=_DEV)
def spmm_backward(grad: Tensor, rowptr: Tensor, col: Tensor, value: Optional[Tensor],
                  other: Tensor, out: Tensor, reduce: str, need_other: bool,
                  need_value: bool) -> Tuple[Tensor, Tensor]:
    n_rows = rowptr.numel() - 1
    g2 = grad.reshape(n_rows, -1).contiguous()
    o2 = _rows(other)
    g_other = other.new_empty(0)
    g_value = other.new_empty(0)
    if reduce in ('min', 'max'):
        if need_other:
            g_other = _native.spmm_minmax_backward_dst(rowptr, col, o2, out.reshape(n_rows, -1),
                                                       g2, other.size(0)).reshape(other.shape)
        return g_other, g_value
    if reduce == 'mean':
        inv = 1.0 / (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(torch.float32)
        g2 = g2 * inv.view(-1, 1)
    if need_other:
        # edge-parallel transposed product on the CSR's own slots (no second sort): atomics
        dst = _native.ptr2index(rowptr, col.numel())
        g_other = _native.gather_scatter_add(g2, dst, col, other.size(0),
                                             w=value).reshape(other.shape)
    if need_value and value is not None:
        g_value = _native.sddmm_csr(rowptr, col, None, g2, o2, col.numel(), 1).reshape(value.shape)
    return g_other, g_value


@spmm_backward.register_fake
def _(grad, rowptr, col, value, other, out, reduce, need_other, need_value):
    g_other = torch.empty_like(other) if need_other else other.new_empty(0)
    g_value = (torch.empty_like(value) if (need_value and value is not None)
               else other.new_empty(0))
    return g_other, g_value


def _spmm_setup(ctx, inputs, output):
    rowptr, col, value, other, reduce = inputs
    ctx.reduce, ctx.has_value = reduce, value is not None
    ctx.save_for_backward(rowptr, col, value, other, output)


def _spmm_bwd(ctx, grad):
    rowptr, col, value, other, out = ctx.saved_tensors
    need_other, need_value = ctx.needs_input_grad[3], ctx.needs_input_grad[2] and ctx.has_value
    g_other, g_value = spmm_backward(grad.contiguous(), rowptr, col, value, other, out,
                                     ctx.reduce, need_other, need_value)
    return None, None, (g_value if need_value else None), (g_other if need_other else None), None


register_autograd('pyg_amd::spmm', _spmm_bwd, setup_context=_spmm_setup)


# ---- dense transform ----------------------------------------------------------------------------------
@custom_op('pyg_amd::linear', mutates_args=(), device_types=_DEV)
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    out = _native.linear_forward(x.reshape(-1, x.size(-1)), weight, bias)
    return out.reshape(*x.shape[:-1], weight.size(0))


@linear.register_fake
def _(x, weight, bias=None):
    return x.new_empty(*x.shape[:-1], weight.size(0))


@custom_op('pyg_amd::linear_backward', mutates_args=(), device_types=_DEV)
def linear_backward(grad: Tensor, x: Tensor, weight: Tensor, need_x: bool, need_w: bool,
                    need_b: bool) -> Tuple[Tensor, Tensor, Tensor]:
    g2 = grad.reshape(-1, grad.size(-1)).contiguous()
    x2 = x.reshape(-1, x.size(-1))
    empty = x.new_empty(0)
    gx = (_native.linear_dgrad(g2, weight.t().contiguous()).reshape(x.shape) if need_x
          else empty)
    gw = gb = empty
    if need_w:
        gw = _native.linear_wgrad(g2, x2, bias_grad=need_b)
        if need_b:
            gw, gb = gw
    elif need_b:
        gb = _native.colsum(g2)
    return gx, gw, gb


@linear_backward.register_fake
def _(grad, x, weight, need_x, need_w, need_b):
    e = x.new_empty(0)
    return (torch.empty_like(x) if need_x else e, torch.empty_like(weight) if need_w else e,
            x.new_empty(weight.size(0)) if need_b else e)


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.has_bias = bias is not None
    ctx.save_for_backward(x, weight)


def _linear_bwd(ctx, grad):
    x, weight = ctx.saved_tensors
    nx, nw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    nb = ctx.has_bias and ctx.needs_input_grad[2]
    gx, gw, gb = linear_backward(grad.contiguous(), x, weight, nx, nw, nb)
    return (gx if nx else None), (gw if nw else None), (gb if nb else None)


register_autograd('pyg_amd::linear', _linear_bwd, setup_context=_linear_setup)

OPS = ('index_sort', 'index2ptr', 'ptr2index', 'gather', 'scatter', 'scatter_backward',
       'segment_csr', 'segment_csr_backward', 'softmax_csr', 'softmax_csr_backward', 'spmm',
       'spmm_backward', 'linear', 'linear_backward')

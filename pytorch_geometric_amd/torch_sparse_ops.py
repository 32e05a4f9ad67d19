"""``torch.ops.torch_sparse.spmm_{sum,mean,min,max}`` — the operator names and argument lists the
reference dispatches to for a sparse-dense product on the GPU
(``torch_geometric/edge_index.py:1798-1810``, ``_torch_sparse_spmm``), served by this backend when
the ``torch-sparse`` extension itself is absent (seam S2 of SURVEY.md §8(b)): a reference whose
``torch_geometric.typing.WITH_TORCH_SPARSE`` is switched on finds the MI355X kernels behind the
symbols it already calls, without ``backend.install()`` rebinding anything.

Schemas (those of torch-sparse's ``csrc/spmm.cpp``):

* ``spmm_sum(Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr,
  Tensor? csr2csc, Tensor mat) -> Tensor``
* ``spmm_mean(Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? rowcount,
  Tensor? colptr, Tensor? csr2csc, Tensor mat) -> Tensor``
* ``spmm_min / spmm_max(Tensor rowptr, Tensor col, Tensor? value, Tensor mat) -> (Tensor, Tensor)``

``colptr`` / ``csr2csc`` (the transposed pointer and the CSR -> CSC permutation the reference passes
when ``mat`` needs a gradient) turn the backward into an SpMM over the transposed form — no atomics;
without them the edge-parallel atomic kernel runs.  Device tensors only: there is no CPU kernel
behind these names (a CPU tensor raises, as everywhere in this package)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _native

_LIB = None
_SCHEMAS = {
    'spmm_sum': '(Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, '
                'Tensor? csr2csc, Tensor mat) -> Tensor',
    'spmm_mean': '(Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? rowcount, '
                 'Tensor? colptr, Tensor? csr2csc, Tensor mat) -> Tensor',
    'spmm_min': '(Tensor rowptr, Tensor col, Tensor? value, Tensor mat) -> (Tensor, Tensor)',
    'spmm_max': '(Tensor rowptr, Tensor col, Tensor? value, Tensor mat) -> (Tensor, Tensor)',
}


def _rows(t: Tensor) -> Tensor:
    return t.reshape(t.size(0), -1)


class _SpmmSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rowptr, col, value, colptr, csr2csc, row, mat, reduce):
        _native._require_device(rowptr, col, mat)
        n_rows = rowptr.numel() - 1
        out = _native.spmm_csr(rowptr, col, _rows(mat), reduce, n_rows=n_rows, w=value,
                               hub=_native.hub_plan(rowptr))
        ctx.reduce, ctx.mat_shape = reduce, mat.shape
        ctx.save_for_backward(rowptr, col, value, colptr, csr2csc, row, mat)
        return out.reshape(n_rows, *mat.shape[1:])

    @staticmethod
    def backward(ctx, grad):
        rowptr, col, value, colptr, csr2csc, row, mat = ctx.saved_tensors
        n_rows = rowptr.numel() - 1
        g2 = grad.reshape(n_rows, -1).contiguous()
        if ctx.reduce == 'mean':
            inv = 1.0 / (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(torch.float32)
            g2 = g2 * inv.view(-1, 1)
        g_mat = g_val = None
        if ctx.needs_input_grad[6]:
            if colptr is not None and csr2csc is not None:
                # transposed product over the CSC form the caller already holds: entry e of column
                # c is CSR slot csr2csc[e], whose row is the source of the transposed edge
                r = row if row is not None else _native.ptr2index(rowptr, col.numel())
                g_mat = _native.spmm_csr(colptr, r[csr2csc], g2, 'sum', n_rows=mat.size(0),
                                         w=None if value is None else value[csr2csc],
                                         hub=_native.hub_plan(colptr))
            else:
                dst = row if row is not None else _native.ptr2index(rowptr, col.numel())
                g_mat = _native.gather_scatter_add(g2, dst, col, mat.size(0), w=value)
            g_mat = g_mat.reshape(ctx.mat_shape)
        if value is not None and ctx.needs_input_grad[2]:
            g_val = _native.sddmm_csr(rowptr, col, None, g2, _rows(mat), col.numel(),
                                      1).reshape(value.shape)
        return None, None, g_val, None, None, None, g_mat, None


class _SpmmMinMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rowptr, col, value, mat, reduce):
        _native._require_device(rowptr, col, mat)
        if value is not None:
            raise NotImplementedError('edge weights are not supported for min / max aggregation '
                                      '(the reference never passes them: edge_index.py:1846-1851)')
        n_rows = rowptr.numel() - 1
        out, arg = _native.spmm_csr(rowptr, col, _rows(mat), reduce, n_rows=n_rows,
                                    hub=_native.hub_plan(rowptr), return_arg=True)
        ctx.save_for_backward(rowptr, col, mat, out)
        ctx.mat_shape = mat.shape
        ctx.mark_non_differentiable(arg)
        return out.reshape(n_rows, *mat.shape[1:]), arg.reshape(n_rows, *mat.shape[1:])

    @staticmethod
    def backward(ctx, grad, _grad_arg):
        rowptr, col, mat, out = ctx.saved_tensors
        n_rows = rowptr.numel() - 1
        g = _native.spmm_minmax_backward_dst(rowptr, col, _rows(mat), out,
                                             grad.reshape(n_rows, -1).contiguous(), mat.size(0))
        return None, None, None, g.reshape(ctx.mat_shape), None


def _spmm_sum(row, rowptr, col, value, colptr, csr2csc, mat):
    return _SpmmSumMean.apply(rowptr, col, value, colptr, csr2csc, row, mat, 'sum')


def _spmm_mean(row, rowptr, col, value, rowcount, colptr, csr2csc, mat):
    return _SpmmSumMean.apply(rowptr, col, value, colptr, csr2csc, row, mat, 'mean')


def _spmm_min(rowptr, col, value, mat) -> Tuple[Tensor, Tensor]:
    return _SpmmMinMax.apply(rowptr, col, value, mat, 'min')


def _spmm_max(rowptr, col, value, mat) -> Tuple[Tensor, Tensor]:
    return _SpmmMinMax.apply(rowptr, col, value, mat, 'max')


def torch_sparse_present() -> bool:
    import importlib.util
    return importlib.util.find_spec('torch_sparse') is not None


def register(force: bool = False) -> Optional[bool]:
    """Define ``torch_sparse::spmm_{sum,mean,min,max}`` and bind them to this backend.  Returns
    ``True`` when (already) registered by this module, ``False`` when the real ``torch-sparse`` is
    installed (its own kernels keep the names; pass ``force=True`` only in a test process that
    never imports it)."""
    global _LIB
    if _LIB is not None:
        return True
    if torch_sparse_present() and not force:
        return False
    lib = torch.library.Library('torch_sparse', 'DEF')
    impls = {'spmm_sum': _spmm_sum, 'spmm_mean': _spmm_mean, 'spmm_min': _spmm_min,
             'spmm_max': _spmm_max}
    for name, schema in _SCHEMAS.items():
        lib.define(name + schema)
        # composite: the body is an autograd.Function over the C-ABI kernels, so the names work
        # with and without gradients (no CPU fallback: _require_device raises on a CPU tensor)
        lib.impl(name, impls[name], 'CompositeImplicitAutograd')
    _LIB = lib
    return True

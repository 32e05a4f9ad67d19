"""Tensor-level wrappers over the C ABI: validate, allocate outputs with the torch caching
allocator, pass raw device pointers + the current HIP stream.  No arithmetic happens here.

PyTorch is plumbing only (device memory, streams).  Every function requires HIP device tensors
and raises otherwise — there is no CPU fallback.
"""
import collections
import ctypes
import os
import threading
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _compiled, _lib
from ._lib import REDUCE_IDS, PygAmdError, SpmmArgs, check

# rows with more stored entries than this are split into chunks (see csrc/spmm.hip)
HUB_THRESHOLD = 1024
# (256-slot chunks: a chunk is ONE wave's chain of gathers — at 1,024 slots the ~700 chunks of the
# products shape were 700 waves on 1,024 SIMDs, 151 us per 256-wide launch; at 256: 71 us.  128
# gains nothing more: profiles/r06_bench_kernel_stats.md)
HUB_CHUNK = int(os.environ.get('PYGAMD_HUB_CHUNK', '256'))


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PygAmdError(
                'pytorch_geometric_amd kernels need HIP device tensors (got a '
                f'{t.device} tensor); there is no CPU fallback on this path')


def _idx_dtype(t: Tensor) -> int:
    if t.dtype == torch.int64:
        return _lib.IDX_I64
    if t.dtype == torch.int32:
        return _lib.IDX_I32
    raise ValueError(f"index tensors must be int32 or int64 (got {t.dtype})")


def _p(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(ref: Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)


def _f32_rows(t: Tensor, name: str) -> Tensor:
    """2-D fp32 view with unit inner stride (row stride may exceed the width)."""
    if t.dtype != torch.float32:
        raise ValueError(f"'{name}' must be float32 (got {t.dtype})")
    if t.dim() != 2:
        raise ValueError(f"'{name}' must be two-dimensional (got {t.dim()} dimensions)")
    if t.size(1) > 0 and t.size(0) > 0 and (t.stride(1) != 1 or t.stride(0) < t.size(1)):
        t = t.contiguous()
    return t


def _ld(t: Tensor) -> int:
    return t.stride(0) if (t.size(0) > 1 and t.size(1) > 0) else max(t.size(1), 1)


def _plain(*tensors) -> bool:
    """Every given tensor is a 2-D fp32 HIP tensor: the operands the compiled binding
    (csrc/torch_binding.cpp, `torch.ops.pyg_amd_c`) takes as they are.  Anything else goes through
    the Python path below, which raises this package's typed errors."""
    for t in tensors:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
            return False
    return True


def _hub4(hub):
    """(rows, chunk_ptr, n_hub, n_chunks) -> the same with Nones for 'no hub rows'"""
    if hub is not None and hub[2] > 0:
        return hub
    return None, None, 0, 0


# ---- integer side --------------------------------------------------------------------------------
def index_sort(keys: Tensor, max_value: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    _require_device(keys)
    if keys.dim() != 1:
        raise ValueError("'inputs' must be one-dimensional")
    keys = keys.contiguous()
    lib = _lib.load()
    n = keys.numel()
    out = torch.empty_like(keys)
    perm = torch.empty(n, dtype=torch.int64, device=keys.device)
    if n == 0:
        return out, perm
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_index_sort_workspace_bytes(_idx_dtype(keys), n, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=keys.device)
    check(lib.pygamd_index_sort(_p(keys), _idx_dtype(keys), n,
                                -1 if max_value is None else int(max_value), _p(out), _p(perm),
                                _p(ws), nbytes.value, _stream(keys)), 'index_sort')
    return out, perm


def index2ptr(index: Tensor, size: int) -> Tensor:
    _require_device(index)
    C = _compiled.ops()
    if C is not None and index.dtype in (torch.int32, torch.int64):
        return C.index2ptr(index, size)
    index = index.contiguous()
    lib = _lib.load()
    ptr = torch.empty(size + 1, dtype=index.dtype, device=index.device)
    check(lib.pygamd_index2ptr(_p(index), _idx_dtype(index), index.numel(), size, _p(ptr),
                               _stream(index)), 'index2ptr')
    return ptr


def ptr2index(ptr: Tensor, n: int) -> Tensor:
    _require_device(ptr)
    C = _compiled.ops()
    if C is not None and ptr.dtype in (torch.int32, torch.int64):
        return C.ptr2index(ptr, n)
    ptr = ptr.contiguous()
    lib = _lib.load()
    out = torch.empty(n, dtype=ptr.dtype, device=ptr.device)
    check(lib.pygamd_ptr2index(_p(ptr), _idx_dtype(ptr), ptr.numel() - 1, n, _p(out),
                               _stream(ptr)), 'ptr2index')
    return out


def index_minmax(index: Tensor) -> Tuple[int, int]:
    """(min, max) of an index tensor; one host sync (the reference syncs here too)."""
    _require_device(index)
    index = index.contiguous()
    lib = _lib.load()
    mm = torch.empty(2, dtype=torch.int64, device=index.device)
    check(lib.pygamd_index_minmax(_p(index), _idx_dtype(index), index.numel(), _p(mm),
                                  _stream(index)), 'index_minmax')
    lo, hi = mm.tolist()
    return lo, hi


def permute_index(src: Tensor, perm: Tensor) -> Tensor:
    _require_device(src, perm)
    src = src.contiguous()
    lib = _lib.load()
    out = torch.empty(perm.numel(), dtype=src.dtype, device=src.device)
    check(lib.pygamd_permute_index(_p(src), _idx_dtype(src), _p(perm), perm.numel(), _p(out),
                                   _stream(src)), 'permute_index')
    return out


def cast_index(src: Tensor, dtype: torch.dtype) -> Tensor:
    _require_device(src)
    if src.dtype == dtype:
        return src
    assert src.dtype == torch.int64
    lib = _lib.load()
    out = torch.empty(src.numel(), dtype=dtype, device=src.device)
    check(lib.pygamd_cast_index(_p(src), src.numel(), _idx_dtype(out), _p(out), _stream(src)),
          'cast_index')
    return out


def index_guard(index: Tensor, size: int, err: Optional[Tensor] = None,
                dtype: Optional[torch.dtype] = None) -> Tensor:
    """A copy of ``index`` (as ``dtype``, default its own) with every entry outside ``[0, size)``
    replaced by the sentinel ``size``; ``err`` (device int32[1]) is set when there was one.  No
    host read."""
    _require_device(index, err)
    index = index.contiguous()
    out = torch.empty(index.numel(), dtype=dtype or index.dtype, device=index.device)
    lib = _lib.load()
    check(lib.pygamd_index_guard(_p(index), _idx_dtype(index), index.numel(), int(size), _p(out),
                                 _idx_dtype(out), _p(err), _stream(index)), 'index_guard')
    return out


def cumsum(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """``torch.cumsum(x, 0)`` of a 1-D int32 / int64 device tensor on the own scan kernels
    (``pygamd_cumsum``; ``out`` may be ``x`` or a contiguous slice such as ``offsets[1:]``)."""
    _require_device(x, out)
    if x.dim() != 1 or x.dtype not in (torch.int32, torch.int64):
        raise ValueError(f"'x' must be a one-dimensional int32 / int64 tensor (got {x.dtype}, "
                         f"{x.dim()} dimensions)")
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError("'out' must be contiguous with the shape and dtype of 'x'")
    n = x.numel()
    if n == 0:
        return out
    lib = _lib.load()
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_cumsum_workspace_bytes(_idx_dtype(x), n, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
    check(lib.pygamd_cumsum(_p(x), _idx_dtype(x), n, _p(out), _p(ws), nbytes.value, _stream(x)),
          'cumsum')
    return out


def hub_plan(rowptr: Tensor, threshold: int = None, chunk: int = None):
    """Returns (hub_rows, hub_chunk_ptr, n_hub, n_chunks); tensors are None when n_hub == 0."""
    _require_device(rowptr)
    threshold = HUB_THRESHOLD if threshold is None else threshold
    chunk = HUB_CHUNK if chunk is None else chunk
    lib = _lib.load()
    n_rows = rowptr.numel() - 1
    if n_rows <= 0:
        return None, None, 0, 0
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_hub_plan_workspace_bytes(_idx_dtype(rowptr), n_rows, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=rowptr.device)
    rows = torch.empty(n_rows, dtype=rowptr.dtype, device=rowptr.device)
    cptr = torch.empty(n_rows + 1, dtype=rowptr.dtype, device=rowptr.device)
    n_hub, n_chunks = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.pygamd_hub_plan(_p(rowptr), _idx_dtype(rowptr), n_rows, threshold, chunk, _p(rows),
                              _p(cptr), n_rows, ctypes.byref(n_hub), ctypes.byref(n_chunks),
                              _p(ws), nbytes.value, _stream(rowptr)), 'hub_plan')
    if n_hub.value == 0:
        return None, None, 0, 0
    return (rows[:n_hub.value].clone(), cptr[:n_hub.value + 1].clone(), n_hub.value,
            n_chunks.value)


# ---- CSR SpMM --------------------------------------------------------------------------------------
# Optional launch timing (bench.py): when a list is installed here, every spmm_csr call appends
# (info, start_event, end_event) recorded on the launch stream around the kernel(s).
timing_sink = None


class _timed:
    """Context manager: when bench.py installed a sink, record HIP events on the launch stream
    around the call(s) inside and append ``(info, start, end)``."""

    def __init__(self, info: dict, ref: Tensor):
        self.sink, self.info, self.ref = timing_sink, info, ref

    def __enter__(self):
        if self.sink is not None:
            st = torch.cuda.current_stream(self.ref.device)
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev1 = torch.cuda.Event(enable_timing=True)
            self.ev0.record(st)
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.sink is not None and exc_type is None:
            self.ev1.record(torch.cuda.current_stream(self.ref.device))
            self.sink.append((self.info, self.ev0, self.ev1))
        return False


def spmm_csr(rowptr: Tensor, col: Optional[Tensor], x: Tensor, reduce: str, *,
             n_rows: Optional[int] = None, eid: Optional[Tensor] = None,
             w: Optional[Tensor] = None, src_scale: Optional[Tensor] = None,
             hub=None, out: Optional[Tensor] = None, return_arg: bool = False,
             accumulate: bool = False, hub_phase: int = 0, save_arg32: bool = False,
             relu_mask: Optional[Tensor] = None, relu_bits: Optional[Tensor] = None,
             src_bits: Optional[Tensor] = None, src_bits_set: Optional[Tensor] = None,
             compressed_width: Optional[int] = None, rowend: Optional[Tensor] = None,
             accumulate_rows: int = 0):
    """out[i] = reduce_k m(k) * x[col[k]] — see pygamd_spmm_csr in include/pyg_amd.h.
    ``rowend`` ([n_rows], dtype of ``rowptr``): row ``r`` owns the slots ``[rowptr[r], rowend[r])``
    (fixed-stride slot blocks of a static-shape sampled batch; ``rowptr`` then has ``n_rows``
    entries).  ``accumulate_rows``: with ``accumulate``, only rows below it have an old value.
    ``compressed_width=F``: ``x`` is the int32 block of :func:`rows_compress` holding ``F`` columns
    per row (sum / mean only; the same sums bit for bit).
    ``src_bits`` (from :func:`rows_pack`): one bit per row of ``x``, clear = the row is all zero
    and is not read; ``src_bits_set``: the device counter of set bits (dense inputs then ignore the
    bits)."""
    _require_device(rowptr, col, x, eid, w, src_scale, relu_mask, relu_bits, src_bits,
                    src_bits_set)
    if relu_mask is not None and relu_bits is not None:
        raise ValueError("pass at most one of 'relu_mask' / 'relu_bits'")
    if src_bits is not None:
        if (src_bits.dtype != torch.int32 or not src_bits.is_contiguous()
                or src_bits.numel() < (x.size(0) + 31) // 32):
            raise ValueError(f"'src_bits' must be contiguous int32 with one bit per row of x "
                             f'({(x.size(0) + 31) // 32} words)')
        if src_bits_set is not None and (src_bits_set.dtype != torch.int64
                                         or src_bits_set.numel() != 1):
            raise ValueError("'src_bits_set' must be one int64")
    if compressed_width is not None:
        return _spmm_csr_compressed(rowptr, col, x, reduce, compressed_width, n_rows, hub, out,
                                    hub_phase)
    if rowend is not None:
        _require_device(rowend)
        if rowend.dtype != rowptr.dtype or not rowend.is_contiguous() or n_rows is None \
                or rowend.numel() < n_rows or rowptr.numel() < n_rows:
            raise ValueError("'rowend' needs 'n_rows', the dtype of 'rowptr' and one entry per row")
    C = _compiled.ops()
    if (C is not None and not return_arg and src_bits is None and _plain(x, relu_mask)
            and rowend is None and accumulate_rows == 0
            and (w is None or (w.dtype == torch.float32 and
                               (w.dim() == 1 or x.size(1) % max(w.size(1), 1) == 0)))
            and rowptr.dtype in (torch.int32, torch.int64)):
        nr = rowptr.numel() - 1 if n_rows is None else n_rows
        F = x.size(1)
        if relu_mask is None or tuple(relu_mask.shape) == (nr, F):
            _check_bits(relu_bits, nr, F)
            if out is None:
                out = torch.empty(nr, F, dtype=torch.float32, device=x.device)
            arg32 = None
            if save_arg32 and reduce in ('min', 'max'):
                arg32 = torch.empty(nr, F, dtype=torch.int32, device=x.device)
            h_rows, h_cptr, n_hub, n_chunks = _hub4(hub)
            with _timed({'n_rows': nr, 'n_src': x.size(0),
                         'nnz': col.numel() if col is not None else x.size(0), 'F': F,
                         'reduce': reduce, 'idx_bytes': rowptr.element_size(),
                         'weighted': w is not None, 'src_scale': src_scale is not None,
                         'accumulate': bool(accumulate), 'relu_mask': relu_mask is not None,
                         'relu_bits': relu_bits is not None, 'n_hub': n_hub}, x):
                C.spmm_csr(rowptr, col, x, REDUCE_IDS[reduce], nr, eid, w, src_scale, h_rows,
                           h_cptr, n_hub, n_chunks, HUB_THRESHOLD, HUB_CHUNK, out, accumulate,
                           hub_phase, arg32, relu_mask, relu_bits)
            return (out, arg32) if save_arg32 else out
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    n_rows = rowptr.numel() - 1 if n_rows is None else n_rows
    F = x2.size(1)
    if out is None:
        out = torch.empty(n_rows, F, dtype=torch.float32, device=x.device)
    red = REDUCE_IDS[reduce]
    a = SpmmArgs()
    a.rowptr = rowptr.data_ptr()
    a.col = 0 if col is None else col.data_ptr()
    a.eid = 0 if eid is None else eid.data_ptr()
    w_heads, head_dim = 1, F
    if w is not None:
        if w.dtype != torch.float32:
            raise ValueError("edge weights must be float32")
        w = w.contiguous()
        w_heads = 1 if w.dim() == 1 else w.size(1)
        if w_heads > 1:
            if F % w_heads != 0:
                raise ValueError('feature width must be divisible by the number of heads')
            head_dim = F // w_heads
        a.w = w.data_ptr()
    if src_scale is not None:
        src_scale = src_scale.contiguous()
        a.src_scale = src_scale.data_ptr()
    a.x = x2.data_ptr()
    a.out = out.data_ptr()
    arg = None
    if return_arg:
        arg = torch.empty(n_rows, F, dtype=rowptr.dtype, device=x.device)
        a.arg_out = arg.data_ptr()
    arg32 = None
    if save_arg32 and red in (_lib.MIN, _lib.MAX):
        arg32 = torch.empty(n_rows, F, dtype=torch.int32, device=x.device)
        a.arg32_out = arg32.data_ptr()
    a.n_rows, a.n_src, a.F = n_rows, x2.size(0), F
    a.ldx, a.ldo = _ld(x2), _ld(out)
    a.idx_dtype, a.reduce = _idx_dtype(rowptr), red
    a.w_heads, a.head_dim = w_heads, head_dim
    a.accumulate = 1 if accumulate else 0
    a.hub_phase = hub_phase
    if rowend is not None:
        a.rowend = rowend.data_ptr()
    a.accumulate_rows = int(accumulate_rows)
    if relu_mask is not None:
        m2 = _f32_rows(relu_mask, 'relu_mask')
        if tuple(m2.shape) != (n_rows, F):
            raise ValueError(f"'relu_mask' must be [{n_rows}, {F}], got {tuple(m2.shape)}")
        a.relu_mask, a.ld_mask = m2.data_ptr(), _ld(m2)
    if relu_bits is not None:
        _check_bits(relu_bits, n_rows, F)
        a.relu_bits, a.ld_bits = relu_bits.data_ptr(), relu_bits.size(1)
    if src_bits is not None:
        a.src_bits = src_bits.data_ptr()
        a.src_bits_set = 0 if src_bits_set is None else src_bits_set.data_ptr()
    ws, ws_bytes = None, 0
    if hub is not None and hub[2] > 0:
        hub_rows, hub_cptr, n_hub, n_chunks = hub
        a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), hub_cptr.data_ptr()
        a.n_hub, a.n_chunks = n_hub, n_chunks
        a.hub_threshold, a.hub_chunk = HUB_THRESHOLD, HUB_CHUNK
        # (min / max take the hub list only to schedule those rows first: no workspace)
        if hub_phase != 1 and red in (_lib.SUM, _lib.MEAN):
            ws_bytes = n_chunks * F * 4
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    sink = timing_sink
    if sink is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(x.device))
    check(lib.pygamd_spmm_csr(ctypes.byref(a), _p(ws), ws_bytes, _stream(x)), 'spmm_csr')
    if sink is not None:
        ev1.record(torch.cuda.current_stream(x.device))
        nnz = (col.numel() if col is not None else x2.size(0))
        sink.append(({'n_rows': n_rows, 'n_src': x2.size(0), 'nnz': nnz, 'F': F,
                      'reduce': reduce, 'idx_bytes': rowptr.element_size(),
                      'weighted': w is not None, 'src_scale': src_scale is not None,
                      'accumulate': bool(accumulate), 'relu_mask': relu_mask is not None,
                      'relu_bits': relu_bits is not None, 'src_bits': src_bits is not None,
                      'n_hub': a.n_hub}, ev0, ev1))
    if save_arg32:
        return out, arg32
    return (out, arg) if return_arg else out


def compressed_pitch(F: int) -> int:
    """Row pitch (32-bit words) of a compressed block of ``F`` columns: 8 mask words + F values + 4
    words of slack for the 16-byte value loads, rounded up to 128-byte lines (288 for F = 256)."""
    return (8 + F + 4 + 31) // 32 * 32


def _check_compressed(z: Tensor, F: int, name: str):
    if (z.dim() != 2 or z.dtype != torch.int32 or (z.size(0) > 1 and z.stride(0) < F + 12)
            or (z.size(1) > 1 and z.stride(1) != 1) or z.size(1) < F + 12):
        raise ValueError(f"'{name}' must be an int32 [n, >= {F + 12}] block of compressed rows")
    if F > 256 or F % 4 != 0:
        raise ValueError('compressed rows hold at most 256 columns, a multiple of 4')


def rows_compress(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """``x`` ([n, F <= 256] float32) as compressed rows ``[8 mask words | kept values]`` — the
    lossless zero-skipping layout the row gathers read (``pygamd_rows_compress``)."""
    _require_device(x, out)
    x2 = _f32_rows(x, 'x')
    n, F = x2.shape
    if F > 256:
        raise ValueError('compressed rows hold at most 256 columns')
    if out is None:
        out = torch.empty(n, compressed_pitch(F), dtype=torch.int32, device=x.device)
    elif (out.dim() != 2 or out.dtype != torch.int32 or out.size(0) != n or out.size(1) < F + 12
          or (out.size(1) > 1 and out.stride(1) != 1)):
        raise ValueError(f"'out' must be int32 [{n}, >= {F + 12}]")
    check(_lib.load().pygamd_rows_compress(_p(x2), _ld(x2), n, F, _p(out), _ld(out), _stream(x)),
          'rows_compress')
    return out


def _spmm_csr_compressed(rowptr, col, z, reduce, F, n_rows, hub, out, hub_phase):
    _require_device(rowptr, col, z, out)
    _check_compressed(z, F, 'x')
    if reduce not in ('sum', 'mean'):
        raise ValueError("compressed rows: reduce must be 'sum' or 'mean'")
    n_rows = rowptr.numel() - 1 if n_rows is None else n_rows
    if out is None:
        out = torch.empty(n_rows, F, dtype=torch.float32, device=z.device)
    a = SpmmArgs()
    a.rowptr = rowptr.data_ptr()
    a.col = 0 if col is None else col.data_ptr()
    a.x, a.out = z.data_ptr(), out.data_ptr()
    a.n_rows, a.n_src, a.F = n_rows, z.size(0), F
    a.ldx, a.ldo = _ld(z), _ld(out)
    a.idx_dtype, a.reduce = _idx_dtype(rowptr), REDUCE_IDS[reduce]
    a.w_heads, a.head_dim = 1, max(F, 1)
    a.hub_phase = hub_phase
    a.x_format = _lib.X_COMPRESSED
    ws, ws_bytes = None, 0
    if hub is not None and hub[2] > 0:
        hub_rows, hub_cptr, n_hub, n_chunks = hub
        a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), hub_cptr.data_ptr()
        a.n_hub, a.n_chunks = n_hub, n_chunks
        a.hub_threshold, a.hub_chunk = HUB_THRESHOLD, HUB_CHUNK
        if hub_phase != 1:
            ws_bytes = n_chunks * F * 4
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=z.device)
    sink = timing_sink
    if sink is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(z.device))
    check(_lib.load().pygamd_spmm_csr(ctypes.byref(a), _p(ws), ws_bytes, _stream(z)), 'spmm_csr')
    if sink is not None:
        ev1.record(torch.cuda.current_stream(z.device))
        sink.append(({'n_rows': n_rows, 'n_src': z.size(0), 'nnz': col.numel(), 'F': F,
                      'reduce': reduce, 'idx_bytes': rowptr.element_size(), 'weighted': False,
                      'src_scale': False, 'accumulate': False, 'compressed_src': True,
                      'n_hub': a.n_hub}, ev0, ev1))
    return out


def rows_pack(g: Tensor, row_scale: Optional[Tensor] = None, *, scaled: Optional[Tensor] = None,
              copy: Optional[Tensor] = None, count: bool = True):
    """(row_bits, n_set) of ``g`` ([n, F], unit column stride): bit i of ``row_bits`` (int32 words)
    = row i has a non-zero entry; ``n_set`` = one device int64 with the number of such rows (None
    with ``count=False``).  In the same pass: ``scaled[:, :F] = g * row_scale[:, None]`` and
    ``copy[:, :F] = g``, both zero-filled up to their own width (``pygamd_rows_pack``)."""
    _require_device(g, row_scale, scaled, copy)
    if g.dim() != 2 or g.dtype != torch.float32 or (g.size(1) > 1 and g.stride(1) != 1):
        raise ValueError("'g' must be a float32 matrix with unit column stride")
    n, F = g.shape
    for name, t in (('scaled', scaled), ('copy', copy)):
        if t is None:
            continue
        if (t.dim() != 2 or t.dtype != torch.float32 or t.size(0) != n or t.size(1) < F
                or (t.size(1) > 1 and t.stride(1) != 1)):
            raise ValueError(f"'{name}' must be float32 [{n}, >= {F}] with unit column stride")
        if t.data_ptr() == g.data_ptr() and n * F > 0:
            raise ValueError(f"'{name}' may not be 'g' itself")
    if row_scale is not None:
        if row_scale.dtype != torch.float32 or row_scale.numel() != n:
            raise ValueError(f"'row_scale' must be {n} float32 values")
        row_scale = row_scale.contiguous()
    bits = torch.empty((n + 31) // 32, dtype=torch.int32, device=g.device)
    n_set = torch.empty(1, dtype=torch.int64, device=g.device) if count else None
    lib = _lib.load()
    check(lib.pygamd_rows_pack(_p(g), _ld(g), n, F, _p(row_scale), _p(scaled),
                               0 if scaled is None else _ld(scaled),
                               0 if scaled is None else scaled.size(1), _p(copy),
                               0 if copy is None else _ld(copy),
                               0 if copy is None else copy.size(1), _p(bits), _p(n_set),
                               _stream(g)), 'rows_pack')
    return bits, n_set


def multi_reduce_csr(rowptr: Tensor, perm: Optional[Tensor], x: Tensor, want):
    """{'sum' | 'pow_sum' | 'min' | 'max': [n_groups, F]} for the requested names in ONE read of the
    rows (group g = rows x[perm[k]], k in [rowptr[g], rowptr[g+1]); ``perm=None``: rows are already
    grouped)."""
    _require_device(rowptr, perm, x)
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    n_rows, F = rowptr.numel() - 1, x2.size(1)
    names = ('sum', 'pow_sum', 'min', 'max')
    outs = {k: torch.empty(n_rows, F, dtype=torch.float32, device=x.device)
            for k in names if k in want}
    if perm is not None:
        perm = perm.contiguous()
        if perm.dtype != rowptr.dtype:
            perm = perm.to(rowptr.dtype)
    check(lib.pygamd_multi_reduce_csr(_p(rowptr), _p(perm), _idx_dtype(rowptr), _p(x2), _ld(x2),
                                      n_rows, F, *[_p(outs.get(k)) for k in names], max(F, 1),
                                      _stream(x)), 'multi_reduce_csr')
    return outs


def colsum(x: Tensor) -> Tensor:
    """x.sum(0) for a 2-D fp32 tensor (bias gradient)."""
    _require_device(x)
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    out = torch.empty(x2.size(1), dtype=torch.float32, device=x.device)
    check(lib.pygamd_colsum(_p(x2), _ld(x2), x2.size(0), x2.size(1), _p(out), _stream(x)),
          'colsum')
    return out


def spmm_minmax_backward_dst(rowptr: Tensor, col: Optional[Tensor], x: Tensor, out: Tensor,
                             grad_out: Tensor, n_src: int, count_self: bool = True,
                             arg32: Optional[Tensor] = None) -> Tensor:
    """Gradient of the min/max aggregation w.r.t. ``x`` (reference tie rule) from the forward's
    destination-sorted handle.  With the forward's ``arg32`` the outputs with a unique extremum
    take the one-atomic fast path and only the marked ones the two-pass kernel."""
    _require_device(rowptr, col, x, out, grad_out, arg32)
    lib = _lib.load()
    x2, o2, g2 = _f32_rows(x, 'x'), _f32_rows(out, 'out'), _f32_rows(grad_out, 'grad_out')
    F = x2.size(1)
    grad_x = torch.empty(n_src, F, dtype=torch.float32, device=x.device)
    if arg32 is None:
        check(lib.pygamd_spmm_csr_minmax_backward_dst(
            _p(rowptr), _p(col), _idx_dtype(rowptr), _p(x2), _ld(x2), _p(o2), _ld(o2), _p(g2),
            _ld(g2), rowptr.numel() - 1, n_src, F, int(count_self), _p(grad_x), _ld(grad_x),
            _stream(x)), 'spmm_minmax_backward_dst')
    else:
        check(lib.pygamd_spmm_csr_minmax_backward_arg(
            _p(rowptr), _p(col), _idx_dtype(rowptr), _p(arg32.contiguous()), _p(x2), _ld(x2),
            _p(o2), _ld(o2), _p(g2), _ld(g2), rowptr.numel() - 1, n_src, F, int(count_self),
            _p(grad_x), _ld(grad_x), _stream(x)), 'spmm_minmax_backward_arg')
    return grad_x


def spmm_minmax_backward_src(fwd, bwd, slot_map: Tensor, x: Tensor, out: Tensor,
                             grad_out: Tensor, arg32: Tensor, count_self: bool = True
                             ) -> Optional[Tensor]:
    """The min/max gradient without the N x F scattered atomics (``pygamd_spmm_csr_minmax_backward_
    src``): ``fwd`` / ``bwd`` = the destination- / source-sorted CSR of the graph, ``slot_map`` =
    ``EdgeIndex.src_slot_to_dst_slot()``.  Returns ``None`` when the layout is not supported
    (``F % 4``, alignment): the caller then takes :func:`spmm_minmax_backward_dst`."""
    _require_device(fwd.ptr, fwd.idx, bwd.ptr, bwd.idx, slot_map, x, out, grad_out, arg32)
    lib = _lib.load()
    x2, o2, g2 = _f32_rows(x, 'x'), _f32_rows(out, 'out'), _f32_rows(grad_out, 'grad_out')
    F, n_src, nnz = x2.size(1), bwd.ptr.numel() - 1, bwd.idx.numel()
    if F % 4 or _ld(g2) % 4 or g2.data_ptr() % 16:
        return None
    grad_x = torch.empty(n_src, F, dtype=torch.float32, device=x.device)
    nbytes = lib.pygamd_minmax_backward_src_workspace_bytes(fwd.ptr.numel() - 1, nnz, F)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    a32 = arg32.contiguous()
    rc = lib.pygamd_spmm_csr_minmax_backward_src(
        _p(fwd.ptr), _p(fwd.idx), _p(bwd.ptr), _p(bwd.idx), _p(slot_map), _idx_dtype(fwd.ptr),
        _p(a32), _p(x2), _ld(x2), _p(o2), _ld(o2), _p(g2), _ld(g2), fwd.ptr.numel() - 1, n_src,
        nnz, F, int(count_self), _p(ws), nbytes, _p(grad_x), _ld(grad_x), _stream(x))
    if rc == 2:  # PYGAMD_ERR_UNSUPPORTED
        return None
    check(rc, 'spmm_minmax_backward_src')
    return grad_x


def bias_act(x: Tensor, bias: Optional[Tensor], relu: bool) -> Tensor:
    """``act(x + bias)`` as a new contiguous ``[n, F]`` tensor in one pass (``x`` may be a
    row-strided view)."""
    _require_device(x, bias)
    x2 = _f32_rows(x, 'x')
    out = torch.empty(x2.size(0), x2.size(1), dtype=torch.float32, device=x.device)
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != x2.size(1):
            raise ValueError(f"'bias' must be float32 with {x2.size(1)} entries")
        bias = bias.contiguous()
    check(_lib.load().pygamd_bias_act(_p(x2), _ld(x2), _p(bias), x2.size(0), x2.size(1),
                                      int(relu), _p(out), _ld(out), _stream(x)), 'bias_act')
    return out


def relu_backward_colsum(grad: Tensor, act: Tensor, want_colsum: bool = True):
    """(grad * (act > 0) as a new contiguous tensor, its column sums | None) in one pass; ``act``
    is the ReLU output.  Inputs may be row-strided views."""
    _require_device(grad, act)
    lib = _lib.load()
    g2, a2 = _f32_rows(grad, 'grad'), _f32_rows(act, 'act')
    if g2.shape != a2.shape:
        raise ValueError(f'shape mismatch: {tuple(g2.shape)} vs {tuple(a2.shape)}')
    out = torch.empty(g2.size(0), g2.size(1), dtype=torch.float32, device=grad.device)
    cs = torch.empty(g2.size(1), dtype=torch.float32, device=grad.device) if want_colsum else None
    check(lib.pygamd_relu_backward_colsum(_p(g2), _ld(g2), _p(a2), _ld(a2), g2.size(0),
                                          g2.size(1), _p(out), _ld(out), _p(cs), _stream(grad)),
          'relu_backward_colsum')
    return out, cs


def spmm_tie_count(rowptr, col, x, out, count_self: bool) -> Tensor:
    _require_device(rowptr, col, x, out)
    lib = _lib.load()
    x2, o2 = _f32_rows(x, 'x'), _f32_rows(out, 'out')
    ntie = torch.empty_like(o2, memory_format=torch.contiguous_format)
    o2 = o2.contiguous()
    check(lib.pygamd_spmm_csr_tie_count(_p(rowptr), _p(col), _idx_dtype(rowptr), _p(x2), _ld(x2),
                                        _p(o2), _ld(o2), o2.size(0), o2.size(1),
                                        1 if count_self else 0, _p(ntie), _stream(x)),
          'spmm_tie_count')
    return ntie


def spmm_minmax_backward(rowptr_t, col_t, x, out, grad_out, ntie) -> Tensor:
    _require_device(rowptr_t, col_t, x, out, grad_out, ntie)
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    o2, g2, n2 = out.contiguous(), grad_out.contiguous(), ntie.contiguous()
    grad_x = torch.empty(x2.size(0), x2.size(1), dtype=torch.float32, device=x.device)
    check(lib.pygamd_spmm_csr_minmax_backward(_p(rowptr_t), _p(col_t), _idx_dtype(rowptr_t),
                                              _p(x2), _ld(x2), _p(o2), _p(g2), _p(n2), _ld(o2),
                                              x2.size(0), x2.size(1), _p(grad_x), _ld(grad_x),
                                              _stream(x)), 'spmm_minmax_backward')
    return grad_x


def sddmm_csr(rowptr, col, eid, grad_out, x, n_edges: int, w_heads: int) -> Tensor:
    _require_device(rowptr, col, eid, grad_out, x)
    C = _compiled.ops()
    if C is not None and _plain(grad_out, x):
        return C.sddmm_csr(rowptr, col, eid, grad_out, x, n_edges, w_heads)
    lib = _lib.load()
    g2, x2 = _f32_rows(grad_out, 'grad_out'), _f32_rows(x, 'x')
    F = x2.size(1)
    grad_w = torch.zeros(n_edges, w_heads, dtype=torch.float32, device=x.device)
    check(lib.pygamd_sddmm_csr(_p(rowptr), _p(col), _p(eid), _idx_dtype(rowptr), _p(g2), _ld(g2),
                               _p(x2), _ld(x2), rowptr.numel() - 1, F, w_heads,
                               F // max(w_heads, 1), _p(grad_w), _stream(x)), 'sddmm_csr')
    return grad_w


def sddmm_spmm_csr(rowptr, col, eid, rows, x, w, n_edges: int, w_heads: int):
    """(grad_w [n_edges, w_heads], agg [n_rows, F]) from ONE gather of ``x[col[k]]``:
    ``grad_w[e(k), h] = <rows[r, head h], x[col[k], head h]>`` and
    ``agg[r] = sum_k w[e(k), head] * x[col[k]]`` (``pygamd_sddmm_spmm_csr``)."""
    _require_device(rowptr, col, eid, rows, x, w)
    lib = _lib.load()
    r2, x2 = _f32_rows(rows, 'rows'), _f32_rows(x, 'x')
    F = x2.size(1)
    if r2.size(1) != F:
        raise ValueError("'rows' and 'x' must have the same width")
    w2 = w.reshape(-1, w_heads).contiguous()
    if w2.dtype != torch.float32 or w2.size(0) < n_edges:
        raise ValueError("'w' must hold one float32 weight per edge and head")
    n_rows = rowptr.numel() - 1
    agg = torch.empty(n_rows, F, dtype=torch.float32, device=x.device)
    # One 64-lane pass of 16-byte lanes covers 256 columns: every head's dot product is then
    # finished inside the wave and stored.  Wider rows, or rows that only take 4-byte lanes
    # (unaligned views), are covered by several lane groups whose partial sums meet in atomic adds
    # on a zeroed buffer.
    one_pass = (F % 4 == 0 and F <= 256 and (F // max(w_heads, 1)) % 4 == 0
                and all(t.data_ptr() % 16 == 0 and _ld(t) % 4 == 0 for t in (r2, x2, agg)))
    alloc = torch.empty if one_pass else torch.zeros
    grad_w = alloc(n_edges, w_heads, dtype=torch.float32, device=x.device)
    check(lib.pygamd_sddmm_spmm_csr(_p(rowptr), _p(col), _p(eid), _idx_dtype(rowptr), _p(r2),
                                    _ld(r2), _p(x2), _ld(x2), n_rows, F, w_heads,
                                    F // max(w_heads, 1), _p(w2), _p(grad_w), _p(agg), _ld(agg),
                                    _stream(x)), 'sddmm_spmm_csr')
    return grad_w, agg


# ---- unsorted gather / scatter ----------------------------------------------------------------
class IndexOutOfRange(PygAmdError, IndexError):
    """An index outside [0, size) reached a scatter/gather kernel (the reference's ATen CPU
    kernels raise 'index ... is out of bounds' at the same place)."""


# How the error flag of the scatter kernels reaches the host:
#   'async' (default)  the flag is copied to pinned host memory behind the kernel and looked at —
#                      without waiting — at the next scatter call or by `check_index_errors()`
#                      (the reference's GPU path reports such indices as an asynchronous device
#                      assert too; a per-call blocking read would serialise host and device on
#                      every scatter / degree / aggregation forward);
#   'sync'             one blocking 4-byte read per call (raises at the call site);
#   'off'              the kernels still skip such rows, nothing is reported.
INDEX_CHECK = os.environ.get('PYGAMD_CHECK_INDEX', 'async')


class _FlagRing:
    """256 device flags + their pinned host mirrors, handed out round-robin per device."""
    SLOTS = 256

    def __init__(self, device):
        self.dev = torch.zeros(self.SLOTS, dtype=torch.int32, device=device)
        self.host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        self.next = 0
        self.pending = collections.deque()  # (slot, event, what, size, index, style)

    def acquire(self) -> int:
        slot = self.next
        self.next = (self.next + 1) % self.SLOTS
        while self.pending and self.pending[0][0] == slot:  # the ring wrapped: wait for the oldest
            self.pending[0][1].synchronize()
            self.poll()
        return slot

    def publish(self, slot: int, what: str, size: int, index: Tensor, on_flag=None):
        """``on_flag``: called (before the error is raised) when this launch turns out flagged —
        lets the owner of a CACHED object built from ``index`` (a sorted-scatter plan) remember
        it, so that later uses of the cache report the same indices again."""
        self.host[slot:slot + 1].copy_(self.dev[slot:slot + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev.device))
        self.pending.append((slot, ev, what, size, index, _error_style.value, on_flag))

    def poll(self, wait: bool = False):
        while self.pending:
            slot, ev, what, size, index, style, on_flag = self.pending[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
            self.pending.popleft()
            if int(self.host[slot]) != 0:
                self.host[slot] = 0
                self.dev[slot:slot + 1].zero_()
                if on_flag is not None:
                    on_flag()
                _raise_out_of_range(index, size, what, style)


_flag_rings = {}


def _flag_ring(device) -> _FlagRing:
    ring = _flag_rings.get(device)
    if ring is None:
        ring = _flag_rings[device] = _FlagRing(device)
    return ring


def check_index_errors():
    """Waits for every scatter launched so far (on any device) and raises `IndexOutOfRange` if one
    of them met an index outside ``[0, dim_size)`` (see ``INDEX_CHECK``)."""
    for ring in list(_flag_rings.values()):
        ring.poll(wait=True)


# 'dim_size' while an `Aggregation.__call__` is running: a flagged launch is then reported the way
# the reference's wrapper reports it (nn/aggr/base.py:131-141), whenever the flag arrives.  Per
# thread (a DataLoader worker thread's aggregation must not restyle another thread's error) and
# recorded per launch by `_FlagRing.publish`.
class _ErrorStyle(threading.local):
    value = None


_error_style = _ErrorStyle()


def set_error_style(style):
    """Sets this thread's report style for flagged launches; returns the previous one."""
    prev, _error_style.value = _error_style.value, style
    return prev


def _index_flag(device, active: bool):
    """(ring, slot, flag tensor or None) for one index-checking launch, per ``INDEX_CHECK``."""
    if not active or INDEX_CHECK == 'off':
        return None, None, None
    if INDEX_CHECK == 'async' and not torch.cuda.is_current_stream_capturing():
        ring = _flag_ring(device)
        ring.poll()  # raises for an EARLIER launch whose flag has arrived meanwhile
        slot = ring.acquire()
        return ring, slot, ring.dev[slot:slot + 1]
    return None, None, torch.zeros(1, dtype=torch.int32, device=device)


def _index_flag_done(ring, slot, err, what: str, size: int, index: Tensor, on_flag=None):
    if ring is not None:
        ring.publish(slot, what, size, index, on_flag)
    elif err is not None and INDEX_CHECK == 'sync':
        try:
            _raise_if_flagged(err, index, size, what)
        except IndexError:
            if on_flag is not None:
                on_flag()
            raise


def poll_index_errors(wait: bool = False, device=None):
    """Looks at the flags that have arrived (``wait=True``: at all of them, blocking) and raises
    for the first flagged launch.  ``device``: only that device's ring (a caller that knows where
    its launch ran does not touch — or wait on — the other GPUs' rings).  Called at the entry of
    the scatter backward and at the end of an ``Aggregation`` call with a caller-supplied
    ``dim_size``."""
    if device is not None:
        ring = _flag_rings.get(torch.device(device))
        if ring is not None:
            ring.poll(wait=wait)
        return
    for ring in list(_flag_rings.values()):
        ring.poll(wait=wait)


def _raise_out_of_range(index: Tensor, size: int, what: str, style=None):
    lo, hi = index_minmax(index)
    if style == 'edge_index':  # the texts of MessagePassing._index_select_safe
        if lo < 0:
            raise IndexError(
                f"Found negative indices in 'edge_index' (got {lo}). Please ensure that all "
                f"indices in 'edge_index' point to valid indices in the interval [0, {size}) in "
                f"your node feature matrix and try again.")
        raise IndexError(
            f"Found indices in 'edge_index' that are larger than {size - 1} (got {hi}). Please "
            f"ensure that all indices in 'edge_index' point to valid indices in the interval "
            f"[0, {size}) in your node feature matrix and try again.")
    if style == 'dim_size' and size <= hi:
        raise ValueError(f"Encountered invalid 'dim_size' (got '{size}' but expected "
                         f">= '{hi + 1}')")
    bad = hi if hi >= size else lo
    raise IndexOutOfRange(f'{what}: index {bad} is out of bounds for dimension 0 with size '
                          f'{size} (indices span [{lo}, {hi}])')


def _raise_if_flagged(err: Tensor, index: Tensor, size: int, what: str):
    """One 4-byte host read of the kernel's error flag ('sync' mode).  Skipped while the stream is
    being captured into a hipGraph (no sync allowed there; the kernels still skip such rows)."""
    if torch.cuda.is_current_stream_capturing():
        return
    if int(err.item()) != 0:
        style = _error_style.value
        _raise_out_of_range(index, size, what, style if style == 'edge_index' else None)


def gather_rows(x: Tensor, index: Tensor, check_bounds=False) -> Tensor:
    """``x[index]`` for float32 rows.  ``check_bounds``: ``False`` (the caller vouches for the
    index), ``True`` (one blocking flag read, IndexError at the call) or ``'edge_index'`` — the
    gather of ``MessagePassing._index_select``: the reference's IndexError texts
    (message_passing.py:269-290), delivered as ``INDEX_CHECK`` says (async flag ring / blocking
    read / not at all); flagged rows are skipped by the kernel in every mode."""
    _require_device(x, index)
    as_edges = check_bounds == 'edge_index'
    if as_edges and INDEX_CHECK == 'off':
        as_edges = check_bounds = False
    C = _compiled.ops()
    if C is not None and not check_bounds and _plain(x) \
            and index.dtype in (torch.int32, torch.int64):
        return C.gather_rows(x, index)
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    index = index.contiguous()
    n, F = index.numel(), x2.size(1)
    out = torch.empty(n, F, dtype=torch.float32, device=x.device)
    ring = slot = None
    if as_edges:
        ring, slot, err = _index_flag(x.device, n > 0 and F > 0)
    else:
        err = torch.zeros(1, dtype=torch.int32, device=x.device) if check_bounds else None
    check(lib.pygamd_gather_rows(_p(x2), _ld(x2), x2.size(0), _p(index), _idx_dtype(index), n, F,
                                 _p(out), _ld(out), _p(err), _stream(x)), 'gather_rows')
    if as_edges:
        prev = set_error_style('edge_index')
        try:
            _index_flag_done(ring, slot, err, 'gather', x2.size(0), index)
        finally:
            set_error_style(prev)
        return out
    if check_bounds and int(err.item()) != 0:
        lo, hi = index_minmax(index)
        raise IndexError(
            f"Found indices in 'edge_index' outside the valid range [0, {x2.size(0) - 1}] "
            f"(got interval [{lo}, {hi}])")
    return out


def scatter_rows(src: Tensor, index: Tensor, dim_size: int, reduce: str,
                 return_count: bool = False):
    _require_device(src, index)
    lib = _lib.load()
    s2 = _f32_rows(src, 'src')
    index = index.contiguous()
    n, F = index.numel(), s2.size(1)
    red = REDUCE_IDS[reduce]
    out = torch.empty(dim_size, F, dtype=torch.float32, device=src.device)
    need_count = red in (_lib.MEAN, _lib.MIN, _lib.MAX)
    count = torch.empty(dim_size, dtype=torch.float32, device=src.device) if need_count else None
    st = _stream(src)
    check(lib.pygamd_scatter_init(_p(out), _ld(out), dim_size, F, red, _p(count), st),
          'scatter_init')
    capturing = torch.cuda.is_current_stream_capturing()
    ring = slot = None
    if INDEX_CHECK == 'async' and not capturing and n > 0 and F > 0:
        ring = _flag_ring(src.device)
        ring.poll()  # raises for an EARLIER launch whose flag has arrived meanwhile
        slot = ring.acquire()
        err = ring.dev[slot:slot + 1]
    else:
        err = torch.zeros(1, dtype=torch.int32, device=src.device)
    check(lib.pygamd_scatter_rows(_p(s2), _ld(s2), _p(index), _idx_dtype(index), n, F, _p(out),
                                  _ld(out), dim_size, red, _p(count), _p(err), st), 'scatter_rows')
    if ring is not None:
        ring.publish(slot, 'scatter', dim_size, index)
    elif INDEX_CHECK == 'sync' and n > 0 and F > 0:
        # before anything is saved for a backward that would index with the same values
        _raise_if_flagged(err, index, dim_size, 'scatter')
    check(lib.pygamd_scatter_finalize(_p(out), _ld(out), dim_size, F, red, _p(count), st),
          'scatter_finalize')
    return (out, count) if return_count else out


def scatter_minmax_backward(src, index, out, grad_out) -> Tensor:
    _require_device(src, index, out, grad_out)
    lib = _lib.load()
    s2 = _f32_rows(src, 'src')
    o2, g2 = out.contiguous(), grad_out.contiguous()
    index = index.contiguous()
    n, F, dim_size = index.numel(), s2.size(1), o2.size(0)
    ntie = torch.empty_like(o2)
    st = _stream(src)
    check(lib.pygamd_scatter_minmax_tie_count(_p(s2), _ld(s2), _p(index), _idx_dtype(index), n, F,
                                              _p(o2), _ld(o2), dim_size, _p(ntie), st),
          'scatter_minmax_tie_count')
    grad_src = torch.empty(n, F, dtype=torch.float32, device=src.device)
    check(lib.pygamd_scatter_minmax_backward(_p(s2), _ld(s2), _p(index), _idx_dtype(index), n, F,
                                             _p(o2), _p(g2), _p(ntie), _ld(o2), dim_size,
                                             _p(grad_src), _ld(grad_src), st),
          'scatter_minmax_backward')
    return grad_src


def scatter_mul_backward(src, index, out, grad_out) -> Tensor:
    """Gradient of scatter(reduce='mul') w.r.t. ``src`` with ATen's zero-count rule."""
    _require_device(src, index, out, grad_out)
    lib = _lib.load()
    s2 = _f32_rows(src, 'src')
    o2, g2 = out.contiguous(), grad_out.contiguous()
    index = index.contiguous()
    n, F, dim_size = index.numel(), s2.size(1), o2.size(0)
    grad_src = torch.empty(n, F, dtype=torch.float32, device=src.device)
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_scatter_mul_backward_workspace_bytes(dim_size, F, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=src.device)
    check(lib.pygamd_scatter_mul_backward(_p(s2), _ld(s2), _p(index), _idx_dtype(index), n, F,
                                          _p(o2), _p(g2), _ld(o2) if dim_size else max(F, 1),
                                          dim_size, _p(grad_src), _ld(grad_src), _p(ws),
                                          nbytes.value, _stream(src)), 'scatter_mul_backward')
    return grad_src


def scatter_argmax(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    _require_device(src, index)
    lib = _lib.load()
    src = src.contiguous()
    index = index.contiguous()
    gmax = torch.empty(dim_size, dtype=torch.float32, device=src.device)
    arg = torch.empty(dim_size, dtype=index.dtype, device=src.device)
    check(lib.pygamd_scatter_argmax(_p(src), _p(index), _idx_dtype(index), index.numel(),
                                    dim_size, _p(gmax), _p(arg), _stream(src)), 'scatter_argmax')
    return arg


# ---- softmax ---------------------------------------------------------------------------------
def segment_softmax_forward(src: Tensor, ptr: Tensor) -> Tensor:
    _require_device(src, ptr)
    C = _compiled.ops()
    if C is not None and _plain(src):
        return C.segment_softmax_forward(src, ptr)
    lib = _lib.load()
    s2 = src.contiguous()
    out = torch.empty_like(s2)
    check(lib.pygamd_segment_softmax_forward(_p(s2), _p(ptr), _idx_dtype(ptr), ptr.numel() - 1,
                                             s2.size(1), _p(out), _stream(src)),
          'segment_softmax_forward')
    return out


def segment_softmax_backward(out: Tensor, grad_out: Tensor, ptr: Tensor) -> Tensor:
    _require_device(out, grad_out, ptr)
    C = _compiled.ops()
    if C is not None and _plain(out, grad_out):
        return C.segment_softmax_backward(out, grad_out, ptr)
    lib = _lib.load()
    o2, g2 = out.contiguous(), grad_out.contiguous()
    grad_src = torch.empty_like(o2)
    check(lib.pygamd_segment_softmax_backward(_p(o2), _p(g2), _p(ptr), _idx_dtype(ptr),
                                              ptr.numel() - 1, o2.size(1), _p(grad_src),
                                              _stream(out)), 'segment_softmax_backward')
    return grad_src


def softmax_index_forward(src: Tensor, index: Tensor, num_groups: int) -> Tensor:
    """Softmax of ``src [n, H]`` within the groups of an unsorted ``index`` (4 launches)."""
    _require_device(src, index)
    lib = _lib.load()
    s2, idx = src.contiguous(), index.contiguous()
    n, H = s2.shape
    out = torch.zeros_like(s2)
    ws = torch.empty(2 * max(num_groups, 1) * max(H, 1), dtype=torch.float32, device=src.device)
    # an index outside [0, num_groups) is skipped by the kernels and reported like a scatter's
    # (the reference's scatter / index_select path raises there, utils/_softmax.py:82-88)
    ring, slot, err = _index_flag(src.device, n > 0 and H > 0)
    check(lib.pygamd_softmax_index_forward(_p(s2), _p(idx), _idx_dtype(idx), n, H, num_groups,
                                           _p(ws), _p(out), _p(err), _stream(src)),
          'softmax_index_forward')
    _index_flag_done(ring, slot, err, 'softmax', num_groups, idx)
    return out


def softmax_index_backward(out: Tensor, grad_out: Tensor, index: Tensor,
                           num_groups: int) -> Tensor:
    _require_device(out, grad_out, index)
    lib = _lib.load()
    o2, g2, idx = out.contiguous(), grad_out.contiguous(), index.contiguous()
    n, H = o2.shape
    grad_src = torch.empty_like(o2)
    ws = torch.empty(max(num_groups, 1) * max(H, 1), dtype=torch.float32, device=out.device)
    check(lib.pygamd_softmax_index_backward(_p(o2), _p(g2), _p(idx), _idx_dtype(idx), n, H,
                                            num_groups, _p(ws), _p(grad_src), _stream(out)),
          'softmax_index_backward')
    return grad_src


def segment_logsumexp_forward(src: Tensor, ptr: Tensor) -> Tensor:
    _require_device(src, ptr)
    lib = _lib.load()
    s2 = src.contiguous()
    out = torch.empty(ptr.numel() - 1, s2.size(1), dtype=torch.float32, device=src.device)
    check(lib.pygamd_segment_logsumexp_forward(_p(s2), _p(ptr), _idx_dtype(ptr), ptr.numel() - 1,
                                               s2.size(1), _p(out), _stream(src)),
          'segment_logsumexp_forward')
    return out


def segment_logsumexp_backward(src: Tensor, out: Tensor, grad_out: Tensor, ptr: Tensor) -> Tensor:
    _require_device(src, out, grad_out, ptr)
    lib = _lib.load()
    s2, o2, g2 = src.contiguous(), out.contiguous(), grad_out.contiguous()
    grad_src = torch.empty_like(s2)
    check(lib.pygamd_segment_logsumexp_backward(_p(s2), _p(o2), _p(g2), _p(ptr), _idx_dtype(ptr),
                                                ptr.numel() - 1, s2.size(1), _p(grad_src),
                                                _stream(src)), 'segment_logsumexp_backward')
    return grad_src


def gat_edge_softmax_forward(rowptr, col, alpha_src, alpha_dst, slope: float) -> Tensor:
    _require_device(rowptr, col, alpha_src, alpha_dst)
    lib = _lib.load()
    a_s, a_d = alpha_src.contiguous(), alpha_dst.contiguous()
    H = a_s.size(1)
    out = torch.empty(col.numel(), H, dtype=torch.float32, device=a_s.device)
    check(lib.pygamd_gat_edge_softmax_forward(_p(rowptr), _p(col), _idx_dtype(rowptr), _p(a_s),
                                              _p(a_d), rowptr.numel() - 1, H, float(slope),
                                              _p(out), _stream(a_s)), 'gat_edge_softmax_forward')
    return out


def gat_edge_softmax_backward(rowptr, col, alpha_src, alpha_dst, alpha, grad_alpha, slope: float):
    _require_device(rowptr, col, alpha_src, alpha_dst, alpha, grad_alpha)
    lib = _lib.load()
    a_s, a_d = alpha_src.contiguous(), alpha_dst.contiguous()
    al, g = alpha.contiguous(), grad_alpha.contiguous()
    H = a_s.size(1)
    g_src = torch.zeros_like(a_s)
    g_dst = torch.zeros_like(a_d)
    check(lib.pygamd_gat_edge_softmax_backward(_p(rowptr), _p(col), _idx_dtype(rowptr), _p(a_s),
                                               _p(a_d), _p(al), _p(g), rowptr.numel() - 1, H,
                                               float(slope), _p(g_src), _p(g_dst),
                                               _stream(a_s)), 'gat_edge_softmax_backward')
    return g_src, g_dst


# ---- dense feature transform (fp32 MFMA GEMM, csrc/gemm.hip) -------------------------------------
def _nt_workspace(lib, M: int, n_out: int, k_red: int, device):
    """Partial-tile slabs of a launch split over its reduction (few row tiles: sampled blocks,
    Cora-sized inputs); ``(None, 0)`` when the rows alone fill the chip."""
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_linear_nt_workspace_bytes(M, n_out, k_red, ctypes.byref(nbytes)))
    if nbytes.value == 0:
        return None, 0
    return torch.empty(nbytes.value, dtype=torch.uint8, device=device), nbytes.value


def linear_forward(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
                   out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """``act(x @ w.T + bias)`` for row-strided fp32 ``x [M, K]``, ``w [N, K]``; ``out`` may be a
    row-strided view (e.g. one half of an ``[agg | x]`` buffer)."""
    _require_device(x, w, bias, out)
    C = _compiled.ops()
    if C is not None and _plain(x, w) and x.size(1) == w.size(1) \
            and (out is None or (out.dtype == torch.float32 and out.shape == (x.size(0), w.size(0))
                                 and (w.size(0) <= 1 or out.stride(1) == 1))):
        if out is None:
            out = torch.empty(x.size(0), w.size(0), dtype=torch.float32, device=x.device)
        with _timed({'kind': 'gemm', 'op': 'forward', 'M': x.size(0), 'N': w.size(0),
                     'K': x.size(1)}, x):
            C.linear_forward(x, w, bias, relu, out, accumulate)
        return out
    lib = _lib.load()
    x2, w2 = _f32_rows(x, 'x'), _f32_rows(w, 'weight')
    M, K = x2.shape
    N = w2.size(0)
    if w2.size(1) != K:
        raise ValueError(f"'x' has {K} columns but 'weight' expects {w2.size(1)}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    elif out.shape != (M, N) or out.dtype != torch.float32 or (N > 1 and out.stride(1) != 1):
        raise ValueError("'out' must be a float32 [M, N] tensor with unit inner stride")
    if bias is not None:
        bias = bias.contiguous()
    ws, ws_bytes = _nt_workspace(lib, M, N, K, x.device)
    with _timed({'kind': 'gemm', 'op': 'forward', 'M': M, 'N': N, 'K': K}, x):
        check(lib.pygamd_linear_forward(_p(x2), _ld(x2), _p(w2), _ld(w2), _p(bias), M, K, N,
                                        int(relu), int(accumulate), _p(out), _ld(out), _p(ws),
                                        ws_bytes, _stream(x)), 'linear_forward')
    return out


def relu_bits_like(n_rows: int, width: int, device) -> Tensor:
    """Storage for a ReLU mask with one bit per element (``pygamd_spmm_args.relu_bits``): int32
    ``[ceil(n_rows / 32), ceil(width / 32), 32]`` — tiles of 32 rows x 32 columns, word
    ``[r >> 5, c >> 5, r & 31]`` holds the 32 columns of row ``r`` in its bits ``c & 31``."""
    return torch.empty((n_rows + 31) // 32, (width + 31) // 32, 32, dtype=torch.int32,
                       device=device)


def pack_relu_bits(act: Tensor) -> Tensor:
    """``act > 0`` in the layout of :func:`relu_bits_like` (host-side helper for tests / callers
    that did not get the bits from ``sage_layer_forward``)."""
    n, f = act.shape
    rt, w = (n + 31) // 32, (f + 31) // 32
    pos = torch.zeros(rt * 32, w * 32, dtype=torch.int64, device=act.device)
    pos[:n, :f] = (act > 0).to(torch.int64)
    weights = (1 << torch.arange(32, dtype=torch.int64, device=act.device))
    words = (pos.view(rt, 32, w, 32) * weights).sum(-1)          # [tile, row in tile, block]
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)  # two's complement int32
    return words.permute(0, 2, 1).contiguous().to(torch.int32)


def _check_bits(bits: Optional[Tensor], n_rows: int, width: int):
    if bits is None:
        return
    if (bits.dtype != torch.int32 or bits.dim() != 3 or bits.size(0) < (n_rows + 31) // 32
            or bits.size(1) < (width + 31) // 32 or bits.size(2) != 32
            or not bits.is_contiguous()):
        raise ValueError(f"'relu_bits' must be a contiguous int32 [>= {(n_rows + 31) // 32}, >= "
                         f"{(width + 31) // 32}, 32] (see relu_bits_like), got {bits.dtype} "
                         f"{tuple(bits.shape)}")


def sage_layer_forward_supported(F: int, Fo: int, reduce: str) -> bool:
    return bool(_lib.load().pygamd_sage_layer_forward_supported(F, Fo, REDUCE_IDS[reduce]))


# schedule of the one-kernel layer: 0 / None = the production entry point (arithmetic per
# set_gemm_mode); 1..6 = include/pyg_amd_lab.h (1 = production fp32 schedule with probe bits,
# 2 = streamed gather phase, 3 / 4 = persistent producer / consumer waves with 4 / 8 transform
# waves, 5 / 6 = the production split / fp32 kernels whatever the mode); the env switch exists for
# A/B timing on the device
SAGE_FUSED_VARIANT = int(os.environ.get('PYGAMD_FUSED_VARIANT', '0'))
SAGE_FUSED_PROBE = 0  # scripts/fused_probe.py: skip the gather (1) / MFMA (2) loop of the kernel


def sage_layer_forward(rowptr: Tensor, col: Tensor, x_gather: Tensor, x_root: Tensor, w: Tensor,
                       bias: Optional[Tensor], reduce: str, relu: bool, agg: Tensor, out: Tensor,
                       hub=None, save_agg: bool = True, hub_threshold: int = None,
                       hub_chunk: int = None, relu_bits: Optional[Tensor] = None,
                       mask_bits: Optional[Tensor] = None, row_scale: Optional[Tensor] = None,
                       out_scaled: Optional[Tensor] = None, variant: Optional[int] = None,
                       gather_width: Optional[int] = None,
                       compressed_out: Optional[Tensor] = None,
                       rowend: Optional[Tensor] = None) -> Tensor:
    """``out = act([aggr(x_gather) | x_root] @ w.T + bias)`` in ONE kernel (csrc/sage_fused.hip);
    ``agg`` ([n_rows, F] view, may be a half of a wider buffer) receives the aggregated rows when
    ``save_agg`` (hub rows always).  ``relu_bits`` (from :func:`relu_bits_like`, needs
    ``relu``) receives ``out > 0`` as one bit per element (see :func:`relu_bits_like`).
    ``mask_bits`` (same layout): ``out`` is zeroed where its bit is clear — with the transposed
    graph, the degree-scaled gradient rows as ``x_gather``, the unscaled ones as ``x_root`` and
    ``w = [W_l^T | W_r^T]`` the launch is the layer's INPUT GRADIENT (dgrad GEMM + transposed
    aggregation + ReLU backward in one pass).  ``out_scaled`` receives ``out * row_scale[:, None]``
    as a second output (what the next such launch gathers).  ``gather_width=F``: ``x_gather`` is
    an int32 block of compressed rows (:func:`rows_compress`) of ``F`` columns; ``compressed_out``
    (int32 ``[n_rows, compressed_pitch(Fo)]``) receives ``out`` once more in that layout."""
    _require_device(rowptr, col, x_gather, x_root, w, bias, agg, out, relu_bits, mask_bits,
                    row_scale, out_scaled, compressed_out)
    variant = SAGE_FUSED_VARIANT if variant is None else variant
    lab = bool(variant or SAGE_FUSED_PROBE)  # a laboratory schedule / probe: libpyg_amd_lab.so
    C = None if lab else _compiled.ops()
    if (C is not None and gather_width is None and compressed_out is None and rowend is None
            and _plain(x_gather, x_root, w, agg, out, out_scaled)):
        n_rows, F, Fo = rowptr.numel() - 1, x_gather.size(1), w.size(0)
        ok = (x_root.shape == (n_rows, F) and w.size(1) == 2 * F and agg.shape == (n_rows, F)
              and out.shape == (n_rows, Fo)
              and (out_scaled is None or (row_scale is not None and row_scale.numel() == n_rows
                                          and row_scale.dtype == torch.float32
                                          and out_scaled.shape == (n_rows, Fo)
                                          and (Fo <= 1 or out_scaled.stride(1) == 1))))
        if ok:
            _check_bits(relu_bits, n_rows, Fo)
            _check_bits(mask_bits, n_rows, Fo)
            h_rows, h_cptr, n_hub, n_chunks = _hub4(hub)
            with _timed({'n_rows': n_rows, 'n_src': x_gather.size(0), 'nnz': col.numel(), 'F': F,
                         'reduce': reduce, 'idx_bytes': rowptr.element_size(), 'weighted': False,
                         'src_scale': False, 'accumulate': False, 'n_hub': n_hub,
                         'fused_gemm': {'Fo': Fo, 'K': 2 * F, 'save_agg': bool(save_agg),
                                        'backward': mask_bits is not None,
                                        'scaled_copy': out_scaled is not None}}, x_gather):
                C.sage_layer_fused(
                    rowptr, col, x_gather, x_root, w, bias, REDUCE_IDS[reduce], relu, agg, out,
                    h_rows, h_cptr, n_hub, n_chunks,
                    HUB_THRESHOLD if hub_threshold is None else hub_threshold,
                    HUB_CHUNK if hub_chunk is None else hub_chunk, save_agg, relu_bits,
                    mask_bits, row_scale, out_scaled)
            return out
    lib = _lib.load_lab() if lab else _lib.load()
    xr, w2 = _f32_rows(x_root, 'x_root'), _f32_rows(w, 'weight')
    if gather_width is None:
        xg = _f32_rows(x_gather, 'x')
        F = xg.size(1)
    else:
        _check_compressed(x_gather, gather_width, 'x_gather')
        xg, F = x_gather, gather_width
    n_rows, Fo = rowptr.numel() - 1, w2.size(0)
    if rowend is not None:  # rows own fixed slot blocks [rowptr[r], rowend[r]) (sampled batches)
        _require_device(rowend)
        n_rows = xr.size(0)
        if rowend.dtype != rowptr.dtype or not rowend.is_contiguous() \
                or rowend.numel() < n_rows or rowptr.numel() < n_rows:
            raise ValueError("'rowend' needs the dtype of 'rowptr' and one entry per row")
    if xr.shape != (n_rows, F) or w2.size(1) != 2 * F or agg.shape != (n_rows, F) \
            or out.shape != (n_rows, Fo):
        raise ValueError('shape mismatch in sage_layer_forward')
    a = SpmmArgs()
    a.rowptr, a.col = rowptr.data_ptr(), col.data_ptr()
    if rowend is not None:
        a.rowend = rowend.data_ptr()
    a.x, a.out = xg.data_ptr(), agg.data_ptr()
    a.n_rows, a.n_src, a.F = n_rows, xg.size(0), F
    a.ldx, a.ldo = _ld(xg), _ld(agg)
    a.idx_dtype, a.reduce = _idx_dtype(rowptr), REDUCE_IDS[reduce]
    a.w_heads, a.head_dim = 1, F
    if gather_width is not None:
        a.x_format = _lib.X_COMPRESSED
    if hub is not None and hub[2] > 0:
        hub_rows, hub_cptr, n_hub, n_chunks = hub
        a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), hub_cptr.data_ptr()
        a.n_hub, a.n_chunks = n_hub, n_chunks
        a.hub_threshold = HUB_THRESHOLD if hub_threshold is None else hub_threshold
        a.hub_chunk = HUB_CHUNK if hub_chunk is None else hub_chunk
    if bias is not None:
        bias = bias.contiguous()
    _check_bits(relu_bits, n_rows, Fo)
    _check_bits(mask_bits, n_rows, Fo)
    f = _lib.SageFusedArgs()
    f.x_root, f.ld_root = xr.data_ptr(), _ld(xr)
    f.w, f.ldw = w2.data_ptr(), _ld(w2)
    f.bias = 0 if bias is None else bias.data_ptr()
    f.Fo, f.relu, f.save_agg = Fo, int(relu), int(save_agg)
    f.y, f.ldy = out.data_ptr(), _ld(out)
    if relu_bits is not None:
        f.relu_bits_out, f.ld_bits_out = relu_bits.data_ptr(), relu_bits.size(1)
    if mask_bits is not None:
        f.mask_bits, f.ld_mask_bits = mask_bits.data_ptr(), mask_bits.size(1)
    if out_scaled is not None:
        if row_scale is None or row_scale.numel() != n_rows or row_scale.dtype != torch.float32:
            raise ValueError("'out_scaled' needs a float32 'row_scale' with one entry per row")
        if out_scaled.shape != (n_rows, Fo) or out_scaled.dtype != torch.float32 \
                or (Fo > 1 and out_scaled.stride(1) != 1):
            raise ValueError("'out_scaled' must be a float32 [n_rows, Fo] tensor with unit inner "
                             "stride")
        row_scale = row_scale.contiguous()
        f.row_scale = row_scale.data_ptr()
        f.y_scaled, f.ldy_scaled = out_scaled.data_ptr(), _ld(out_scaled)
    if compressed_out is not None:
        _check_compressed(compressed_out, Fo, 'compressed_out')
        if compressed_out.size(0) != n_rows or Fo % 32 != 0:
            raise ValueError("'compressed_out' needs one row per output row and Fo % 32 == 0")
        f.compressed_out, f.ld_compressed = compressed_out.data_ptr(), _ld(compressed_out)
    sink = timing_sink
    if sink is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(xg.device))
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_sage_layer_fused_workspace_bytes(ctypes.byref(a), ctypes.byref(f),
                                                      ctypes.byref(nbytes)))
    ws_bytes = nbytes.value
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xg.device) if ws_bytes else None
    if lab:
        check(lib.pygamd_lab_sage_layer_fused(ctypes.byref(a), ctypes.byref(f), variant or 1,
                                              SAGE_FUSED_PROBE, _p(ws), ws_bytes, _stream(xg)),
              'sage_layer_forward (lab schedule)')
    else:
        check(lib.pygamd_sage_layer_fused(ctypes.byref(a), ctypes.byref(f), _p(ws), ws_bytes,
                                          _stream(xg)), 'sage_layer_forward')
    if sink is not None:
        ev1.record(torch.cuda.current_stream(xg.device))
        sink.append(({'n_rows': n_rows, 'n_src': xg.size(0), 'nnz': col.numel(), 'F': F,
                      'reduce': reduce, 'idx_bytes': rowptr.element_size(), 'weighted': False,
                      'src_scale': False, 'accumulate': False, 'n_hub': a.n_hub,
                      'compressed_src': gather_width is not None,
                      'fused_gemm': {'Fo': Fo, 'K': 2 * F, 'save_agg': bool(save_agg),
                                     'backward': mask_bits is not None,
                                     'scaled_copy': out_scaled is not None,
                                     'compressed_out': compressed_out is not None}}, ev0, ev1))
    return out


def cross_entropy_rows(logits: Tensor, target: Tensor, index: Optional[Tensor] = None):
    """``(loss, grad_rows)`` of ``F.cross_entropy(logits[index], target[index])`` (``index`` None: all
    rows; mean reduction) from ONE pass over the selected rows (``pygamd_cross_entropy_step``):
    ``loss`` a float32 scalar, ``grad_rows`` ``[B, C]`` = ``d loss / d logits[index]``.  ``target``
    holds one int64 label per row of ``logits``.  A label outside ``[0, C)`` / an index outside the
    rows is flagged (``PYGAMD_CHECK_INDEX``) and contributes neither loss nor gradient."""
    _require_device(logits, target, index)
    x = _f32_rows(logits, 'logits')
    N, C = x.shape
    if target.dtype != torch.int64 or target.dim() != 1 or target.numel() != N \
            or not target.is_contiguous():
        raise ValueError(f"'target' must hold one contiguous int64 label per row of 'logits' "
                         f"({N}), got {target.dtype} {tuple(target.shape)}")
    if index is not None and (index.dtype != torch.int64 or index.dim() != 1
                              or not index.is_contiguous()):
        raise ValueError("'index' must be a contiguous 1-D int64 tensor")
    B = N if index is None else index.numel()
    if B == 0 or C == 0:
        raise ValueError('cross_entropy_rows needs at least one row and one class')
    lib = _lib.load()
    grad = torch.empty(B, C, dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_cross_entropy_step_workspace_bytes(B, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
    ring, slot, err = _index_flag(x.device, True)
    check(lib.pygamd_cross_entropy_step(_p(x), _ld(x), _p(index), N, B, C, _p(target), _p(index),
                                        _p(grad), C, _p(loss), _p(ws), nbytes.value, _p(err),
                                        None, _stream(x)), 'cross_entropy_rows')
    _index_flag_done(ring, slot, err, "cross_entropy: a target outside [0, C) (or an 'index' "
                     "entry outside the rows of the logits)", C, target)
    return loss, grad


def set_gemm_mode(mode: str) -> str:
    """Arithmetic of the dense transform kernels (``pygamd_set_gemm_mode``): ``'fp32'`` (default:
    the fp32 matrix instruction, bitwise an fmaf chain) or ``'split'`` (operands as three bf16
    terms each, six bf16 matrix products with fp32 accumulation: fp32-level accuracy, ~2.5 x the
    matrix rate).  Returns the previous mode."""
    lib = _lib.load()
    if mode not in _lib.GEMM_MODES:
        raise ValueError(f"mode must be one of {sorted(_lib.GEMM_MODES)}, got '{mode}'")
    prev = get_gemm_mode()
    for each in _lib.loaded():  # (the laboratory build keeps its own copy of the switch)
        check(each.pygamd_set_gemm_mode(_lib.GEMM_MODES[mode]), 'set_gemm_mode')
    return prev


def lab_set_wgrad_variant(variant: int) -> None:
    """LAB (include/pyg_amd_lab.h, libpyg_amd_lab.so): 0 = production split weight gradient
    (operands split once on their way into LDS), 1 = round 3's in-register schedule, further
    values = timing probes.  While a variant is selected EVERY call of this module goes to the
    laboratory library (a superset build of the product library) through ctypes; 0 switches back
    to the product library.  Tests and probes only."""
    check(_lib.load_lab().pygamd_lab_set_wgrad_variant(int(variant)), 'lab_set_wgrad_variant')
    _lib.use_lab(int(variant) != 0)


def get_gemm_mode() -> str:
    code = _lib.load().pygamd_get_gemm_mode()
    return next(k for k, v in _lib.GEMM_MODES.items() if v == code)


def linear_dgrad(g: Tensor, w_t: Tensor, row_scale: Optional[Tensor] = None, n_scaled: int = 0,
                 out: Optional[Tensor] = None, accumulate: bool = False,
                 relu_mask: Optional[Tensor] = None, relu_bits: Optional[Tensor] = None,
                 out_scaled: Optional[Tensor] = None) -> Tensor:
    """``g [M, N] @ w [N, K]`` with the weight handed over transposed (``w_t [K, N]``); columns
    ``[0, n_scaled)`` of the result are multiplied by ``row_scale[row]``; where ``relu_mask [M, K]``
    (a ReLU output) is not positive the result is 0 (that ReLU's backward as the epilogue);
    ``relu_bits`` is the same mask as one bit per element (:func:`relu_bits_like`).
    ``out_scaled [M, K]`` receives ``result * row_scale[:, None]`` (all columns) as a second
    output of the same pass."""
    _require_device(g, w_t, row_scale, out, relu_mask, relu_bits, out_scaled)
    if relu_mask is not None and relu_bits is not None:
        raise ValueError("pass at most one of 'relu_mask' / 'relu_bits'")
    C = _compiled.ops()
    if C is not None and _plain(g, w_t, relu_mask, out, out_scaled) \
            and g.size(1) == w_t.size(1):
        M, K = g.size(0), w_t.size(0)
        ok = ((out is None or (out.shape == (M, K) and (K <= 1 or out.stride(1) == 1)))
              and (relu_mask is None or tuple(relu_mask.shape) == (M, K))
              and (out_scaled is None or (row_scale is not None and not accumulate
                                          and out_scaled.shape == (M, K)
                                          and (K <= 1 or out_scaled.stride(1) == 1))))
        if ok:
            _check_bits(relu_bits, M, K)
            if out is None:
                out = torch.empty(M, K, dtype=torch.float32, device=g.device)
            with _timed({'kind': 'gemm', 'op': 'dgrad', 'M': M, 'N': K, 'K': g.size(1)}, g):
                C.linear_dgrad(g, w_t, row_scale, n_scaled, out, accumulate, relu_mask,
                               relu_bits, out_scaled)
            return out
    lib = _lib.load()
    g2, w2 = _f32_rows(g, 'grad'), _f32_rows(w_t, 'weight_t')
    M, N = g2.shape
    K = w2.size(0)
    if w2.size(1) != N:
        raise ValueError(f"'grad' has {N} columns but 'weight_t' expects {w2.size(1)}")
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=g.device)
    if row_scale is not None:
        row_scale = row_scale.contiguous()
    if out_scaled is not None:
        if row_scale is None or accumulate:
            raise ValueError("'out_scaled' needs 'row_scale' and no 'accumulate'")
        if out_scaled.shape != (M, K) or out_scaled.dtype != torch.float32 \
                or (K > 1 and out_scaled.stride(1) != 1):
            raise ValueError("'out_scaled' must be a float32 [M, K] tensor with unit inner stride")
    m2 = None
    _check_bits(relu_bits, M, K)
    if relu_mask is not None:
        m2 = _f32_rows(relu_mask, 'relu_mask')
        if tuple(m2.shape) != (M, K):
            raise ValueError(f"'relu_mask' must be [{M}, {K}], got {tuple(m2.shape)}")
    ws, ws_bytes = _nt_workspace(lib, M, K, N, g.device)
    with _timed({'kind': 'gemm', 'op': 'dgrad', 'M': M, 'N': K, 'K': N}, g):
        check(lib.pygamd_linear_dgrad2(_p(g2), _ld(g2), _p(w2), _ld(w2), _p(row_scale),
                                       n_scaled if row_scale is not None else 0, M, N, K,
                                       int(accumulate), _p(m2), _ld(m2) if m2 is not None else 0,
                                       _p(relu_bits),
                                       relu_bits.size(1) if relu_bits is not None else 0,
                                       _p(out), _ld(out), _p(out_scaled),
                                       _ld(out_scaled) if out_scaled is not None else 0,
                                       _p(ws), ws_bytes, _stream(g)), 'linear_dgrad')
    return out


def linear_wgrad(g: Tensor, x: Tensor, out: Optional[Tensor] = None,
                 accumulate: bool = False, wgs_per_cu: int = 0, bias_grad: bool = False,
                 x2: Optional[Tensor] = None, bias_out: Optional[Tensor] = None):
    """``g [M, N].T @ x [M, K]`` -> ``[N, K]`` (deterministic split reduction over M).
    ``wgs_per_cu=1`` halves the launch's footprint (for running under a bandwidth-bound kernel
    on another stream).  ``bias_grad=True`` also returns ``g.sum(0)`` — taken from the same pass
    over ``g`` — as ``(grad_w, grad_b)``.  ``x2`` (``[M, K2]``): the gradient against
    ``[x | x2]`` -> ``[N, K + K2]`` without concatenating the two.  ``bias_out`` (contiguous
    float32 ``[N]``, with ``bias_grad``): where the bias gradient is written (e.g. its slot of a
    flat gradient buffer) instead of a fresh tensor."""
    _require_device(g, x, out, x2, bias_out)
    if bias_out is not None and (not bias_grad or bias_out.dtype != torch.float32
                                 or bias_out.shape != (g.size(1), )
                                 or not bias_out.is_contiguous()):
        raise ValueError("'bias_out' needs bias_grad=True and a contiguous float32 [N] tensor")
    C = _compiled.ops()
    if C is not None and _plain(g, x, x2, out) and x.size(0) == g.size(0) \
            and (x2 is None or (x2.size(0) == g.size(0) and x.size(1) > 0 and x2.size(1) > 0)):
        N, K = g.size(1), x.size(1) + (0 if x2 is None else x2.size(1))
        if K > 0 and (out is None or (out.shape == (N, K) and (K <= 1 or out.stride(1) == 1))):
            if out is None:
                out = torch.empty(N, K, dtype=torch.float32, device=g.device)
            gb = None
            if bias_grad:
                gb = bias_out if bias_out is not None else \
                    torch.empty(N, dtype=torch.float32, device=g.device)
            with _timed({'kind': 'gemm', 'op': 'wgrad', 'M': g.size(0), 'N': N, 'K': K,
                         'x2': x2 is not None}, g):
                C.linear_wgrad(g, x, out, accumulate, wgs_per_cu, gb, x2)
            return (out, gb) if bias_grad else out
    lib = _lib.load()
    second = None if x2 is None else _f32_rows(x2, 'x2')
    g2, first = _f32_rows(g, 'grad'), _f32_rows(x, 'x')
    M, N = g2.shape
    K1 = first.size(1)
    K2 = 0 if second is None else second.size(1)
    K = K1 + K2
    if first.size(0) != M or (second is not None and second.size(0) != M):
        raise ValueError(f"'grad' has {M} rows but 'x' has {first.size(0)}"
                         + ('' if second is None else f" and 'x2' {second.size(0)}"))
    if second is not None and (K1 == 0 or K2 == 0):
        raise ValueError("both operands of a two-operand weight gradient need columns")
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=g.device)
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_linear_wgrad_workspace_bytes(M, N, K, ctypes.byref(nbytes)))
    ws = torch.empty(max(nbytes.value, 4), dtype=torch.uint8, device=g.device)
    gb = None
    if bias_grad:
        if K == 0:  # no weight tile passes over g
            cs = colsum(g2)
            return out, (cs if bias_out is None else bias_out.copy_(cs))
        gb = bias_out if bias_out is not None else \
            torch.empty(N, dtype=torch.float32, device=g.device)
    with _timed({'kind': 'gemm', 'op': 'wgrad', 'M': M, 'N': N, 'K': K,
                 'x2': second is not None}, g):
        check(lib.pygamd_linear_wgrad2(_p(g2), _ld(g2), _p(first), _ld(first), K1, _p(second),
                                       _ld(second) if second is not None else 0, K2, M, N,
                                       int(accumulate), int(wgs_per_cu), _p(out), _ld(out),
                                       _p(gb), _p(ws), nbytes.value, _stream(g)), 'linear_wgrad')
    return (out, gb) if bias_grad else out


# ---- segment_matmul (grouped GEMM, fp32 MFMA) ---------------------------------------------------
_segmm_plans = {}


WGRAD_CHUNK_ROWS = 2048


def segmm_plan(ptr_host: tuple, device, blocks: int = 1):
    """(tiles int32 [T,3], T, wgrad chunks int32 [C,3], C) on `device` for a host pointer tuple;
    cached.  Tiles cover every non-empty segment in 64-row pieces, chunks in 2048-row pieces.
    With ``blocks`` > 1 every piece is listed once per column block, as group
    ``segment * blocks + b`` (the block-diagonal layout of ``pygamd_segment_matmul``)."""
    key = (ptr_host, str(device), blocks)
    hit = _segmm_plans.get(key)
    if hit is not None:
        return hit
    tm = _lib.load().pygamd_segment_matmul_tile_rows()

    def pieces(step):
        out = []
        for g in range(len(ptr_host) - 1):
            a, b = ptr_host[g], ptr_host[g + 1]
            for r in range(a, b, step):
                for blk in range(blocks):
                    out.append((g * blocks + blk, r, min(step, b - r)))
        return out

    tiles, chunks = pieces(tm), pieces(WGRAD_CHUNK_ROWS)
    t = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 3).to(device)
    c = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 3).to(device)
    if len(_segmm_plans) > 32:
        _segmm_plans.pop(next(iter(_segmm_plans)))
    _segmm_plans[key] = (t, len(tiles), c, len(chunks))
    return _segmm_plans[key]


def _row_index(rows: Optional[Tensor], name: str) -> Optional[Tensor]:
    if rows is None:
        return None
    if rows.dim() != 1 or rows.dtype not in (torch.int32, torch.int64):
        raise ValueError(f"'{name}' must be a one-dimensional int32 / int64 tensor")
    return rows.to(torch.int64).contiguous()


def segment_matmul(x: Tensor, w: Tensor, plan, transpose_w: bool = False,
                   blocks: int = 1, x_rows: Optional[Tensor] = None) -> Tensor:
    """out[seg] = x[seg] @ W[g]  (or @ W[g]^T when transpose_w) — W is [G, K, N] contiguous.
    ``blocks`` > 1: x is [S, blocks*K], W is [segments*blocks, K, N], out is [S, blocks*N].
    ``x_rows`` [S]: operand row s is ``x[x_rows[s]]`` (gathered inside the kernel)."""
    _require_device(x, w, x_rows)
    x_rows = _row_index(x_rows, 'x_rows')
    lib = _lib.load()
    tiles, n_tiles = plan[0], plan[1]
    x2 = _f32_rows(x, 'x')
    w = w.contiguous()
    if transpose_w:
        w = w.transpose(1, 2)
    G, K, N = w.shape
    # the bf16 term planes of the weights (split arithmetic, K <= 128: csrc/segmm.hip)
    nbytes = ctypes.c_size_t(0)
    check(lib.pygamd_segment_matmul_workspace_bytes(G, K, N, ctypes.byref(nbytes)))
    split = nbytes.value > 0 and get_gemm_mode() == 'split'
    if transpose_w and not split:
        # the general kernel streams weight rows with coalesced 128-byte loads: materialise W^T
        # (the split kernel's pre-pass reads the transposed view where it lies)
        w = w.contiguous()
    seg_stride, sk, sn = w.stride()
    if x2.size(1) != blocks * K:
        raise ValueError(f"'inputs' has {x2.size(1)} columns but the weights expect "
                         f"{blocks * K}")
    n_out = x2.size(0) if x_rows is None else x_rows.numel()
    out = torch.empty(n_out, blocks * N, dtype=torch.float32, device=x.device)
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device) if split else None
    check(lib.pygamd_segment_matmul(_p(x2), _ld(x2), _p(x_rows), _p(w), seg_stride, sk, sn, G,
                                    _p(tiles),
                                    n_tiles, K, N, blocks, _p(out), _ld(out), _p(ws),
                                    0 if ws is None else nbytes.value, _stream(x)),
          'segment_matmul')
    return out


def segment_matmul_wgrad(x: Tensor, g: Tensor, plan, n_seg: int, blocks: int = 1,
                         g_rows: Optional[Tensor] = None) -> Tensor:
    """grad_W[g] = x[seg]^T @ grad[seg] -> [G, K, N]  (G = n_seg groups incl. blocks).
    ``g_rows`` [S]: gradient row s is ``g[g_rows[s]]`` (gathered inside the kernel)."""
    _require_device(x, g, g_rows)
    g_rows = _row_index(g_rows, 'g_rows')
    lib = _lib.load()
    x2, g2 = _f32_rows(x, 'x'), _f32_rows(g, 'grad')
    K, N = x2.size(1) // blocks, g2.size(1) // blocks
    gw = torch.empty(n_seg, K, N, dtype=torch.float32, device=x.device)
    check(lib.pygamd_segment_matmul_wgrad(_p(x2), _ld(x2), _p(g2), _ld(g2), _p(g_rows),
                                          _p(plan[2]), plan[3],
                                          n_seg, K, N, blocks, _p(gw), _stream(x)),
          'segment_matmul_wgrad')
    return gw


# ---- neighbour sampling (one hop) ---------------------------------------------------------------
def sample_neighbors(colptr: Tensor, row: Tensor, frontier: Tensor, offsets: Tensor, total: int,
                     max_per_node: int, seed: int, zero_fill: bool = False,
                     replace: bool = False, salt_position: bool = False,
                     seed_dev: Optional[Tensor] = None):
    """(src_global, dst_pos_in_frontier, csc_slot) for the sampled in-edges of `frontier`.
    ``total`` sizes the outputs; ``zero_fill`` for a static capacity larger than what the hop
    really samples (the tail then holds 0 = a valid node / slot id).  ``replace``: draws with
    replacement (``offsets`` from :func:`sample_counts` with the same flag).  ``salt_position``:
    the draws of a node also depend on its position in ``frontier`` (disjoint sampling: the same
    node in two trees draws independently).  ``seed_dev`` (int64 [1] on the device) is added to
    ``seed`` on the device (a captured graph bumps it between replays)."""
    _require_device(colptr, row, frontier, offsets)
    lib = _lib.load()
    alloc = torch.zeros if zero_fill else torch.empty
    src = alloc(total, dtype=colptr.dtype, device=colptr.device)
    dstpos = alloc(total, dtype=colptr.dtype, device=colptr.device)
    slot = alloc(total, dtype=colptr.dtype, device=colptr.device)
    if total > 0:
        check(lib.pygamd_sample_neighbors(_p(colptr), _p(row), _idx_dtype(colptr), _p(frontier),
                                          frontier.numel(), _p(offsets), max_per_node,
                                          seed & 0xFFFFFFFFFFFFFFFF,
                                          int(replace) | (2 if salt_position else 0),
                                          _p(seed_dev), _p(src),
                                          _p(dstpos),
                                          _p(slot), _stream(colptr)), 'sample_neighbors')
    return src, dstpos, slot


def gather_scatter_add(x: Tensor, gather_idx: Tensor, scatter_idx: Tensor, n_out: int,
                       scale: Optional[Tensor] = None, w: Optional[Tensor] = None,
                       out: Optional[Tensor] = None, n_valid: Optional[Tensor] = None) -> Tensor:
    """out[scatter_idx[e]] += scale[gather_idx[e]] * w[e] * x[gather_idx[e]]; ``out`` defaults to
    zeros, or accumulates into the given (row-strided) buffer.  ``n_valid`` (int64 [1], device):
    only that many leading entries of the fixed-capacity edge list are real."""
    _require_device(x, gather_idx, scatter_idx, scale, w, out)
    C = _compiled.ops()
    if C is not None and _plain(x, out) and gather_idx.dtype in (torch.int32, torch.int64) \
            and (out is None or (out.size(1) == x.size(1) and (x.size(1) <= 1 or out.stride(1) == 1))):
        if out is None:
            out = torch.zeros(n_out, x.size(1), dtype=torch.float32, device=x.device)
        C.gather_scatter_add(x, gather_idx, scatter_idx, scale, w, out, n_valid)
        return out
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    F = x2.size(1)
    if out is None:
        out = torch.zeros(n_out, F, dtype=torch.float32, device=x.device)
    gi, si = gather_idx.contiguous(), scatter_idx.contiguous()
    if scale is not None:
        scale = scale.contiguous()
    if w is not None:
        w = w.contiguous()
    check(lib.pygamd_gather_scatter_add(_p(x2), _ld(x2), _p(gi), _p(si), _idx_dtype(gi),
                                        _p(scale), _p(w), gi.numel(), _p(n_valid), F, _p(out),
                                        _ld(out), _stream(x)), 'gather_scatter_add')
    return out


# ---- GAT node terms ----------------------------------------------------------------------------
def head_dot_forward(x: Tensor, att_a: Tensor, att_b: Optional[Tensor], H: int, C: int):
    """(a, b) with a[n,h] = <x[n,h,:], att_a[h,:]>; x is [N, H*C]."""
    _require_device(x, att_a, att_b)
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    n = x2.size(0)
    att_a = att_a.contiguous()
    out_a = torch.empty(n, H, dtype=torch.float32, device=x.device)
    out_b = None
    if att_b is not None:
        att_b = att_b.contiguous()
        out_b = torch.empty(n, H, dtype=torch.float32, device=x.device)
    check(lib.pygamd_head_dot_forward(_p(x2), _ld(x2), _p(att_a), _p(att_b), n, H, C, _p(out_a),
                                      _p(out_b), _stream(x)), 'head_dot_forward')
    return out_a, out_b


def head_dot_backward(x: Tensor, att_a: Tensor, att_b: Optional[Tensor], grad_a: Tensor,
                      grad_b: Optional[Tensor], H: int, C: int, need_grad_x: bool,
                      accumulate_into: Optional[Tensor] = None):
    """``accumulate_into`` ([n, H*C], unit inner stride): the input gradient is ADDED to it (the
    gradient of the same x through the aggregation) instead of going to a fresh tensor."""
    _require_device(x, att_a, att_b, grad_a, grad_b, accumulate_into)
    lib = _lib.load()
    x2 = _f32_rows(x, 'x')
    n = x2.size(0)
    att_a, grad_a = att_a.contiguous(), grad_a.contiguous()
    g_att_a = torch.empty(H * C, dtype=torch.float32, device=x.device)
    g_att_b = None
    if att_b is not None:
        att_b, grad_b = att_b.contiguous(), grad_b.contiguous()
        g_att_b = torch.empty(H * C, dtype=torch.float32, device=x.device)
    if accumulate_into is not None:
        grad_x = _f32_rows(accumulate_into, 'accumulate_into')
        if grad_x is not accumulate_into or grad_x.shape != (n, H * C):
            raise ValueError("'accumulate_into' must be [n, H * C] float32 with unit inner stride")
    else:
        grad_x = torch.empty(n, H * C, dtype=torch.float32, device=x.device) \
            if need_grad_x else None
    check(lib.pygamd_head_dot_backward(_p(x2), _ld(x2), _p(att_a), _p(att_b), _p(grad_a),
                                       _p(grad_b), n, H, C, _p(grad_x),
                                       _ld(grad_x) if grad_x is not None else 0,
                                       1 if accumulate_into is not None else 0, _p(g_att_a),
                                       _p(g_att_b), _stream(x)), 'head_dot_backward')
    return grad_x, g_att_a, g_att_b


def sample_counts(colptr: Tensor, frontier: Tensor, k: int,
                  n_valid: Optional[Tensor] = None, replace: bool = False) -> Tensor:
    """min(deg, k) per frontier entry (``replace`` with ``k >= 0``: k wherever deg > 0); entries
    past the device-side count ``n_valid`` (int64 [1]) of a fixed-capacity frontier get 0."""
    _require_device(colptr, frontier, n_valid)
    lib = _lib.load()
    cnt = torch.empty_like(frontier)
    check(lib.pygamd_sample_counts(_p(colptr), _idx_dtype(colptr), _p(frontier),
                                   frontier.numel(), k, int(replace and k >= 0), _p(n_valid),
                                   _p(cnt), _stream(colptr)),
          'sample_counts')
    return cnt


def relabel_new_nodes(src_global: Tensor, local_map: Tensor, base: int):
    """Assigns local ids base, base+1, ... to the sources not yet in `local_map` (order of first
    appearance) and returns (new_nodes, row_local).  One host sync (the number of new nodes)."""
    _require_device(src_global, local_map)
    lib = _lib.load()
    m = src_global.numel()
    dt = _idx_dtype(src_global)
    st = _stream(src_global)
    if m == 0:
        return src_global.new_empty(0), src_global.new_empty(0)
    check(lib.pygamd_relabel(0, _p(src_global), dt, m, None, _p(local_map), None, 0, None, None,
                             st))
    flag = torch.empty(m, dtype=torch.int64, device=src_global.device)
    check(lib.pygamd_relabel(1, _p(src_global), dt, m, None, _p(local_map), _p(flag), 0, None,
                             None, st))
    scan = cumsum(flag)
    n_new = int(scan[-1])  # host sync: sizes the next hop (the frontier)
    new_nodes = torch.empty(n_new, dtype=src_global.dtype, device=src_global.device)
    check(lib.pygamd_relabel(2, _p(src_global), dt, m, None, _p(local_map), _p(scan), base, None,
                             _p(new_nodes) if n_new > 0 else _p(flag), st))
    rows = torch.empty_like(src_global)
    check(lib.pygamd_relabel(3, _p(src_global), dt, m, None, _p(local_map), None, 0, None,
                             _p(rows), st))
    return new_nodes, rows


def relabel_new_nodes_padded(src_global: Tensor, total: Tensor, local_map: Tensor, base: Tensor):
    """The same WITHOUT a host sync: ``src_global`` has the hop's static capacity, ``total`` (int64
    [1], device) of its entries are real, ``base`` (int64 [1], device) nodes are in the batch so
    far.  Returns (new_nodes [capacity] zero-padded, row_local [capacity] zero-padded, n_new int64
    [1] on the device)."""
    _require_device(src_global, total, local_map, base)
    lib = _lib.load()
    m = src_global.numel()
    dt = _idx_dtype(src_global)
    st = _stream(src_global)
    dev = src_global.device
    new_nodes = torch.zeros(m, dtype=src_global.dtype, device=dev)
    rows = torch.empty_like(src_global)
    if m == 0:
        return new_nodes, rows, torch.zeros(1, dtype=torch.int64, device=dev)
    check(lib.pygamd_relabel(0, _p(src_global), dt, m, _p(total), _p(local_map), None, 0, None,
                             None, st))
    flag = torch.empty(m, dtype=torch.int64, device=dev)
    check(lib.pygamd_relabel(1, _p(src_global), dt, m, _p(total), _p(local_map), _p(flag), 0,
                             None, None, st))
    scan = cumsum(flag)
    check(lib.pygamd_relabel(2, _p(src_global), dt, m, _p(total), _p(local_map), _p(scan), 0,
                             _p(base), _p(new_nodes), st))
    check(lib.pygamd_relabel(3, _p(src_global), dt, m, _p(total), _p(local_map), None, 0, None,
                             _p(rows), st))
    return new_nodes, rows, scan[-1:]


def edge_key(row: Tensor, col: Tensor, num_nodes: int, by_row: bool) -> Tensor:
    _require_device(row, col)
    lib = _lib.load()
    row, col = row.contiguous(), col.contiguous()
    key = torch.empty(row.numel(), dtype=torch.int64, device=row.device)
    check(lib.pygamd_edge_key(_p(row), _p(col), _idx_dtype(row), row.numel(), num_nodes,
                              int(by_row), _p(key), _stream(row)), 'edge_key')
    return key


def edge_unkey(key_sorted: Tensor, num_nodes: int, by_row: bool, dtype: torch.dtype,
               scan: Optional[Tensor] = None, perm: Optional[Tensor] = None,
               n_out: Optional[int] = None, want_groups: bool = False):
    """(edge_index [2, n_out], gid_orig | None)"""
    _require_device(key_sorted)
    lib = _lib.load()
    E = key_sorted.numel()
    out = torch.empty(2, E if n_out is None else n_out, dtype=dtype, device=key_sorted.device)
    gid = (torch.empty(E, dtype=torch.int64, device=key_sorted.device)
           if (want_groups and scan is not None) else None)
    check(lib.pygamd_edge_unkey(_p(key_sorted), None if scan is None else _p(scan),
                                None if perm is None else _p(perm), E, num_nodes, int(by_row),
                                _idx_dtype(out), _p(out[0]), _p(out[1]),
                                None if gid is None else _p(gid), _stream(key_sorted)),
          'edge_unkey')
    return out, gid


def run_flags(key_sorted: Tensor) -> Tensor:
    _require_device(key_sorted)
    lib = _lib.load()
    flag = torch.empty_like(key_sorted)
    check(lib.pygamd_run_flags(_p(key_sorted), key_sorted.numel(), _p(flag),
                               _stream(key_sorted)), 'run_flags')
    return flag

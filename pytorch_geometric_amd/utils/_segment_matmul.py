import torch
from torch import Tensor
from torch.autograd import Function

from .. import _native


class _SegmentMatmul(Function):
    @staticmethod
    def forward(ctx, inputs: Tensor, ptr_host: tuple, other: Tensor, blocks: int = 1):
        plan = _native.segmm_plan(ptr_host, inputs.device, blocks)
        ctx.plan, ctx.n_seg, ctx.blocks = plan, other.size(0), blocks
        ctx.save_for_backward(inputs, other)
        return _native.segment_matmul(inputs, other, plan, blocks=blocks)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        inputs, other = ctx.saved_tensors
        grad_in = grad_other = None
        if ctx.needs_input_grad[0]:
            grad_in = _native.segment_matmul(grad_out, other, ctx.plan, transpose_w=True,
                                             blocks=ctx.blocks)
        if ctx.needs_input_grad[2]:
            grad_other = _native.segment_matmul_wgrad(inputs, grad_out, ctx.plan, ctx.n_seg,
                                                      ctx.blocks)
        return grad_in, None, grad_other, None


class _SegmentMatmulSum(Function):
    """``segment_matmul`` followed by the per-destination sum of its rows (the tail of RGCNConv,
    rgcn_conv.py:284-290 + aggregate): ``out[i] = sum_{s : dst[s] = i} (inputs[s] @ other[seg(s)])``
    as ONE autograd node.  Forward = the grouped GEMM + one SpMM over ``out_graph`` (rows of the
    product -> destinations).  The backward of that sum gives every row ``s`` the gradient row of
    its destination: instead of gathering them into a ``[S, F]`` copy (one launch, 0.8 GB written
    and read twice at the FB15k-237 shape), the input- and weight-gradient launches read
    ``grad_out[dst[s]]`` themselves (``x_rows`` / ``g_rows`` of the C ABI)."""

    @staticmethod
    def forward(ctx, inputs: Tensor, ptr_host: tuple, other: Tensor, blocks: int, out_graph):
        plan = _native.segmm_plan(ptr_host, inputs.device, blocks)
        t = _native.segment_matmul(inputs, other, plan, blocks=blocks)
        by_dst = out_graph.by_dst()
        out = _native.spmm_csr(by_dst.ptr, by_dst.idx, t, 'sum', n_rows=by_dst.n_rows,
                               hub=by_dst.hub)
        ctx.plan, ctx.n_seg, ctx.blocks = plan, other.size(0), blocks
        ctx.dst = out_graph.edge_index[1]
        ctx.save_for_backward(inputs, other)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        inputs, other = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_in = grad_other = None
        if ctx.needs_input_grad[0]:
            grad_in = _native.segment_matmul(grad_out, other, ctx.plan, transpose_w=True,
                                             blocks=ctx.blocks, x_rows=ctx.dst)
        if ctx.needs_input_grad[2]:
            grad_other = _native.segment_matmul_wgrad(inputs, grad_out, ctx.plan, ctx.n_seg,
                                                      ctx.blocks, g_rows=ctx.dst)
        return grad_in, None, grad_other, None, None


def segment_matmul_sum(inputs: Tensor, ptr_host: tuple, other: Tensor, out_graph) -> Tensor:
    r"""``out[i] = sum_{s: dst[s] = i} inputs[s] @ other[seg(s)]`` with ``out_graph`` the handle
    rows-of-the-product -> destinations (one entry per row ``s``); ``other`` is ``[R, K, N]`` or
    the block-diagonal ``[R, B, K, N]`` of :func:`block_segment_matmul`."""
    if inputs.dtype != torch.float32 or other.dtype != torch.float32:
        raise NotImplementedError("segment_matmul computes in float32 only")
    if other.dim() == 4:
        R, B, K, N = other.shape
        if inputs.size(1) != B * K or len(ptr_host) != R + 1:
            raise ValueError("'inputs' must be [S, B*K] and 'ptr' must hold R + 1 entries")
        return _SegmentMatmulSum.apply(inputs, tuple(ptr_host), other.reshape(R * B, K, N), B,
                                       out_graph)
    if len(ptr_host) != other.size(0) + 1 or inputs.size(1) != other.size(1):
        raise ValueError("'ptr' must hold R + 1 entries and 'inputs' must be [S, K]")
    return _SegmentMatmulSum.apply(inputs, tuple(ptr_host), other, 1, out_graph)


def block_segment_matmul(inputs: Tensor, ptr_host: tuple, other: Tensor) -> Tensor:
    r"""Block-diagonal variant used by ``RGCNConv(num_blocks=B)`` (rgcn_conv.py:222-244):
    ``other`` is ``[R, B, K, N]``; rows of segment ``r`` are multiplied block by block,
    ``out[s, b*N:(b+1)*N] = inputs[s, b*K:(b+1)*K] @ other[r, b]`` — one launch, no transposes."""
    R, B, K, N = other.shape
    if inputs.size(1) != B * K or len(ptr_host) != R + 1:
        raise ValueError("'inputs' must be [S, B*K] and 'ptr' must hold R + 1 entries")
    return _SegmentMatmul.apply(inputs, tuple(ptr_host), other.reshape(R * B, K, N), B)


def segment_matmul(inputs: Tensor, ptr, other: Tensor) -> Tensor:
    r"""``out[ptr[g]:ptr[g+1]] = inputs[ptr[g]:ptr[g+1]] @ other[g]`` — the contract of
    ``pyg_lib.ops.segment_matmul`` (call sites: torch_geometric/nn/conv/rgcn_conv.py:288,
    nn/dense/linear.py:255).  ``ptr`` may be a tensor (one host copy) or a Python sequence.

    ONE launch of the fp32-MFMA grouped GEMM (csrc/segmm.hip) for the forward, one for each
    gradient; the (segment, row tile) table is built once per pointer and cached."""
    if inputs.dtype != torch.float32 or other.dtype != torch.float32:
        raise NotImplementedError("segment_matmul computes in float32 only")
    if inputs.dim() != 2 or other.dim() != 3:
        raise ValueError("'inputs' must be [S, K] and 'other' [G, K, N]")
    if isinstance(ptr, Tensor):
        ptr_host = tuple(int(v) for v in ptr.tolist())
    else:
        ptr_host = tuple(int(v) for v in ptr)
    if len(ptr_host) != other.size(0) + 1:
        raise ValueError(f"'ptr' has {len(ptr_host)} entries but 'other' holds "
                         f"{other.size(0)} matrices")
    if ptr_host[-1] != inputs.size(0) or ptr_host[0] != 0:
        raise ValueError("'ptr' must start at 0 and end at the number of input rows")
    return _SegmentMatmul.apply(inputs, ptr_host, other)

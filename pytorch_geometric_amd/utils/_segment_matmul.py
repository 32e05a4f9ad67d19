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

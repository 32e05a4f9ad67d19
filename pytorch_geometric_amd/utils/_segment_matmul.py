from typing import List

import torch
from torch import Tensor
from torch.autograd import Function


class _SegmentMatmul(Function):
    @staticmethod
    def forward(ctx, inputs: Tensor, ptr_host: tuple, other: Tensor):
        out = inputs.new_empty(inputs.size(0), other.size(-1))
        for r in range(other.size(0)):
            a, b = ptr_host[r], ptr_host[r + 1]
            if b > a:
                torch.mm(inputs[a:b], other[r], out=out[a:b])
        ctx.ptr_host = ptr_host
        ctx.save_for_backward(inputs, other)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        inputs, other = ctx.saved_tensors
        ptr_host = ctx.ptr_host
        grad_in = grad_other = None
        if ctx.needs_input_grad[0]:
            grad_in = torch.empty_like(inputs)
        if ctx.needs_input_grad[2]:
            grad_other = torch.zeros_like(other)
        g = grad_out.contiguous()
        for r in range(other.size(0)):
            a, b = ptr_host[r], ptr_host[r + 1]
            if b <= a:
                continue
            if grad_in is not None:
                torch.mm(g[a:b], other[r].t(), out=grad_in[a:b])
            if grad_other is not None:
                torch.mm(inputs[a:b].t(), g[a:b], out=grad_other[r])
        return grad_in, None, grad_other


def segment_matmul(inputs: Tensor, ptr: Tensor, other: Tensor) -> Tensor:
    r"""``out[ptr[r]:ptr[r+1]] = inputs[ptr[r]:ptr[r+1]] @ other[r]`` — the contract of
    ``pyg_lib.ops.segment_matmul`` (call sites: torch_geometric/nn/conv/rgcn_conv.py:288,
    nn/dense/linear.py:255).  ``ptr`` may be a tensor (one host copy) or a Python sequence.

    Round-1 implementation: one library GEMM (rocBLAS/hipBLASLt, fp32 MFMA) per non-empty
    segment; a single-launch grouped MFMA kernel is the planned replacement."""
    if isinstance(ptr, Tensor):
        ptr_host = tuple(int(v) for v in ptr.tolist())
    else:
        ptr_host = tuple(int(v) for v in ptr)
    if len(ptr_host) != other.size(0) + 1:
        raise ValueError(f"'ptr' has {len(ptr_host)} entries but 'other' holds "
                         f"{other.size(0)} matrices")
    if ptr_host[-1] != inputs.size(0):
        raise ValueError("'ptr[-1]' must equal the number of input rows")
    return _SegmentMatmul.apply(inputs, ptr_host, other)

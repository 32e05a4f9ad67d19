"""Plain data parallelism for mini-batch / replica training: one process per GPU, seed nodes (or
whole synthetic graphs) sharded across ranks, the ONLY exchange step is the gradient all-reduce
(examples/multi_gpu/distributed_sampling.py:64-115 in the reference, where DDP does it).

A 3-layer GraphSAGE has ~0.2-0.3 M parameters (~1 MB): the all-reduce is latency-bound on xGMI,
so all gradients live in ONE flat buffer and each step issues ONE collective (RCCL through
``torch.distributed``'s 'nccl' backend on ROCm, 'gloo' on CPU in the tests)."""
import math

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGradBucket:
    r"""Backs every ``p.grad`` of ``module`` by a view into one contiguous buffer so that the
    whole model is reduced by a single ``all_reduce``."""

    def __init__(self, module: torch.nn.Module, process_group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('module has no trainable parameters')
        p0 = self.params[0]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self.group = process_group
        offset = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[offset:offset + n].view_as(p)
            offset += n

    @property
    def world_size(self) -> int:
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def zero_(self):
        """Use instead of ``optimizer.zero_grad()`` (which would drop the views)."""
        self.flat.zero_()

    def check_views(self) -> bool:
        lo = self.flat.data_ptr()
        hi = lo + self.flat.numel() * self.flat.element_size()
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.params)

    def all_reduce_mean(self, async_op: bool = False, force: bool = False):
        """Average the gradients over all ranks (DDP's AVG semantics): one collective.  A group of
        one rank needs none; ``force`` issues it anyway (exercises the collective path)."""
        ws = self.world_size
        if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
            return None
        self.flat.div_(ws)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=async_op)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """Make every rank start from rank ``src``'s weights (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        flat = torch.cat([p.reshape(-1) for p in module.parameters()])
        dist.broadcast(flat, src=src, group=process_group)
        offset = 0
        for p in module.parameters():
            n = p.numel()
            p.copy_(flat[offset:offset + n].view_as(p))
            offset += n


def shard_seeds(index: Tensor, rank: int, world_size: int) -> Tensor:
    """``train_idx.split(ceil(len / world_size))[rank]``
    (examples/multi_gpu/distributed_sampling.py:70-71)."""
    if world_size <= 1:
        return index
    chunk = math.ceil(index.size(0) / world_size)
    parts = index.split(chunk)
    return parts[rank] if rank < len(parts) else index[:0]

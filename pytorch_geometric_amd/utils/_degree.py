from typing import Optional

import torch
from torch import Tensor

from .. import _native
from .num_nodes import maybe_num_nodes


def degree(index: Tensor, num_nodes: Optional[int] = None,
           dtype: Optional[torch.dtype] = None) -> Tensor:
    r"""Number of occurrences of every node id in :obj:`index`
    (torch_geometric/utils/_degree.py:8-31), computed as a scatter-add of ones."""
    N = maybe_num_nodes(index, num_nodes)
    ones = torch.ones(index.numel(), 1, dtype=torch.float32, device=index.device)
    out = _native.scatter_rows(ones, index, N, 'sum').view(-1)
    return out if dtype is None else out.to(dtype)

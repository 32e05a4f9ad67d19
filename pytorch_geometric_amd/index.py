"""``index2ptr`` / ``ptr2index`` (torch_geometric/index.py:27-37)."""
from typing import Optional

from torch import Tensor

from . import _native


def ptr2index(ptr: Tensor, output_size: Optional[int] = None) -> Tensor:
    if output_size is None:
        output_size = int(ptr[-1]) if ptr.numel() > 0 else 0
    return _native.ptr2index(ptr, output_size)


def index2ptr(index: Tensor, size: Optional[int] = None) -> Tensor:
    if size is None:
        size = _native.index_minmax(index)[1] + 1 if index.numel() > 0 else 0
    return _native.index2ptr(index, size)

from typing import Optional, Tuple

from torch import Tensor

from .. import _native


def index_sort(inputs: Tensor, max_value: Optional[int] = None,
               stable: bool = False) -> Tuple[Tensor, Tensor]:
    r"""Sorts the non-negative integer vector :obj:`inputs` in ascending order and returns
    ``(sorted, perm)`` — same contract as torch_geometric/utils/_index_sort.py:10-32.  The HIP
    radix sort is always stable, so ``perm`` equals ``inputs.sort(stable=True)`` bit for bit
    whether or not ``stable`` is requested."""
    if inputs.dim() != 1:
        raise ValueError(f"'inputs' must be one-dimensional (got {inputs.dim()} dimensions)")
    return _native.index_sort(inputs, max_value)

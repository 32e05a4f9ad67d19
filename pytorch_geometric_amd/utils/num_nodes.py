"""``maybe_num_nodes`` (torch_geometric/utils/num_nodes.py): ``index.max() + 1`` with one host
sync, exactly where the reference has one."""
from typing import Optional

from torch import Tensor

from .. import _native


def maybe_num_nodes(edge_index: Tensor, num_nodes: Optional[int] = None) -> int:
    if num_nodes is not None:
        return num_nodes
    if edge_index.numel() == 0:
        return 0
    if edge_index.is_cuda:
        return _native.index_minmax(edge_index.reshape(-1))[1] + 1
    return int(edge_index.max()) + 1

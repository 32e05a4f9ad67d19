"""Self-loop helpers used by GCNConv / GATConv (torch_geometric/utils/loop.py:71,382,585-657).
Mask / concat bookkeeping on the index tensors; kept as device-agnostic tensor ops exactly like
the reference (SURVEY.md §2.1 'graph utils on the path')."""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from ._scatter import scatter
from .num_nodes import maybe_num_nodes


def contains_self_loops(edge_index: Tensor) -> bool:
    return bool((edge_index[0] == edge_index[1]).any())


def remove_self_loops(edge_index: Tensor,
                      edge_attr: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def _loop_attr(edge_index: Tensor, edge_attr: Tensor, num_nodes: int,
               fill_value: Optional[Union[float, Tensor, str]]) -> Tensor:
    size = (num_nodes, ) + tuple(edge_attr.size()[1:])
    if fill_value is None:
        return edge_attr.new_full(size, 1.0)
    if isinstance(fill_value, (int, float)):
        return edge_attr.new_full(size, float(fill_value))
    if isinstance(fill_value, Tensor):
        attr = fill_value.to(edge_attr.device, edge_attr.dtype)
        if edge_attr.dim() != attr.dim():
            attr = attr.unsqueeze(0)
        return attr.expand(size).contiguous()
    if isinstance(fill_value, str):
        return scatter(edge_attr, edge_index[1], 0, num_nodes, fill_value)
    raise AttributeError("No valid 'fill_value' provided")


def add_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                   fill_value: Optional[Union[float, Tensor, str]] = None,
                   num_nodes: Optional[int] = None) -> Tuple[Tensor, Optional[Tensor]]:
    N = maybe_num_nodes(edge_index, num_nodes)
    loop_index = torch.arange(0, N, dtype=edge_index.dtype, device=edge_index.device)
    loop_index = loop_index.view(1, -1).repeat(2, 1)
    if edge_attr is not None:
        loop_attr = _loop_attr(edge_index, edge_attr, N, fill_value)
        edge_attr = torch.cat([edge_attr, loop_attr], dim=0)
    edge_index = torch.cat([edge_index, loop_index], dim=1)
    return edge_index, edge_attr


def add_remaining_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                             fill_value: Optional[Union[float, Tensor, str]] = None,
                             num_nodes: Optional[int] = None
                             ) -> Tuple[Tensor, Optional[Tensor]]:
    N = maybe_num_nodes(edge_index, num_nodes)
    mask = edge_index[0] != edge_index[1]
    loop_index = torch.arange(0, N, dtype=edge_index.dtype, device=edge_index.device)
    loop_index = loop_index.view(1, -1).repeat(2, 1)
    if edge_attr is not None:
        loop_attr = _loop_attr(edge_index, edge_attr, N, fill_value)
        inv_mask = ~mask
        loop_attr[edge_index[0][inv_mask]] = edge_attr[inv_mask]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    edge_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
    return edge_index, edge_attr

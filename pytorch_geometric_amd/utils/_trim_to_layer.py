from typing import List, Optional, Tuple

from torch import Tensor


def trim_to_layer(layer: int, num_sampled_nodes_per_hop: List[int],
                  num_sampled_edges_per_hop: List[int], x: Tensor, edge_index,
                  edge_attr: Optional[Tensor] = None
                  ) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    r"""Keeps only the node / edge prefix the GNN layer ``layer`` still needs in hop-ordered
    NeighborLoader batches (torch_geometric/utils/_trim_to_layer.py:44-127,167-194): layer ``l > 0``
    drops the last ``num_sampled_*_per_hop[-l]`` nodes and edges."""
    if layer <= 0:
        return x, edge_index, edge_attr
    from ..edge_index import EdgeIndex
    x = x.narrow(0, 0, x.size(0) - num_sampled_nodes_per_hop[-layer])
    if isinstance(edge_index, EdgeIndex):
        num_edges = edge_index.num_edges - num_sampled_edges_per_hop[-layer]
        edge_index = edge_index.trim(x.size(0), num_edges)
        if edge_attr is not None:
            edge_attr = edge_attr.narrow(0, 0, num_edges)
        return x, edge_index, edge_attr
    edge_index = edge_index.narrow(1, 0,
                                   edge_index.size(1) - num_sampled_edges_per_hop[-layer])
    if edge_attr is not None:
        edge_attr = edge_attr.narrow(0, 0,
                                     edge_attr.size(0) - num_sampled_edges_per_hop[-layer])
    return x, edge_index, edge_attr

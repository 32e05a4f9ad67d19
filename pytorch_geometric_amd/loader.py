"""Mini-batch loader over a device-resident graph: the ``NeighborLoader`` role
(torch_geometric/loader/neighbor_loader.py, node_loader.py:90-207, loader/utils.py:32-83,159) for
BASELINE config 4 — seeds are drawn per batch, the k-hop neighbourhood is sampled ON THE GPU
(:mod:`.sampler`), features are gathered with the HIP gather kernel (``filter_data``'s
``x[n_id]``) and the batch never touches the host."""
from dataclasses import dataclass
from typing import Iterator, List, Optional

import torch
from torch import Tensor

from . import _native
from .edge_index import EdgeIndex
from .sampler import NeighborSampler


@dataclass
class Batch:
    """What a model step needs, with the reference's field names (``n_id``, ``e_id``,
    ``batch_size``, ``num_sampled_nodes`` / ``num_sampled_edges`` for ``trim_to_layer``)."""
    x: Tensor
    y: Optional[Tensor]
    edge_index: Tensor
    graph: EdgeIndex       # destination-sorted handle of `edge_index` (no sort, no host sync)
    n_id: Tensor
    e_id: Tensor
    input_id: Tensor
    batch_size: int
    num_sampled_nodes: List[int]
    num_sampled_edges: List[int]


class NeighborLoader:
    r"""Iterates over mini-batches of ``batch_size`` seed nodes with their sampled ``k``-hop
    neighbourhoods.

    Args:
        x, y: node features ``[N, F]`` (fp32, device) and optional labels ``[N]``.
        edge_index: ``[2, E]`` device tensor.
        num_neighbors: fan-out per hop (``-1`` = all).
        input_nodes: seed pool (default: all nodes); shard it across ranks with
            :func:`pytorch_geometric_amd.data_parallel.shard_seeds`.
    """

    def __init__(self, x: Tensor, edge_index: Tensor, num_neighbors: List[int],
                 batch_size: int = 1024, y: Optional[Tensor] = None,
                 input_nodes: Optional[Tensor] = None, shuffle: bool = False,
                 drop_last: bool = False, seed: int = 0):
        self.x, self.y = x, y
        self.num_nodes = x.size(0)
        self.sampler = NeighborSampler(edge_index, self.num_nodes, num_neighbors, seed=seed)
        if input_nodes is None:
            input_nodes = torch.arange(self.num_nodes, device=x.device)
        self.input_nodes = input_nodes.to(x.device)
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self) -> int:
        n = self.input_nodes.numel()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def collate(self, seeds: Tensor, input_id: Optional[Tensor] = None) -> Batch:
        out = self.sampler.sample_from_nodes(seeds)
        x = _native.gather_rows(self.x, out.node)  # filter_data: x[n_id]
        y = None if self.y is None else self.y[out.node]
        ei = torch.stack([out.row, out.col])
        fan = self.sampler.num_neighbors
        graph = EdgeIndex.from_sorted_batch(
            ei, out.node.numel(), max_in_degree=None if min(fan) < 0 else max(fan))
        return Batch(x=x, y=y, edge_index=ei, graph=graph, n_id=out.node,
                     e_id=out.edge, input_id=seeds if input_id is None else input_id,
                     batch_size=seeds.numel(), num_sampled_nodes=out.num_sampled_nodes,
                     num_sampled_edges=out.num_sampled_edges)

    def __iter__(self) -> Iterator[Batch]:
        n = self.input_nodes.numel()
        order = (torch.randperm(n, generator=self._gen).to(self.input_nodes.device)
                 if self.shuffle else torch.arange(n, device=self.input_nodes.device))
        for b in range(len(self)):
            sel = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield self.collate(self.input_nodes[sel], sel)

"""Device-side neighbour sampling (the row that bounds BASELINE config 4).

Mirrors the contract of ``torch_geometric.sampler.NeighborSampler._sample`` ->
``torch.ops.pyg.neighbor_sample`` (torch_geometric/sampler/neighbor_sampler.py:550-620): a CSC
graph, seed nodes and a fan-out per hop in; ``SamplerOutput(node, row, col, edge,
num_sampled_nodes, num_sampled_edges)`` out (torch_geometric/sampler/base.py:168-214) — nodes
ordered seeds first, then hop by hop in order of first appearance (the reference's hash-map
insertion order); edges ordered hop by hop and, inside a hop, by destination;
``row`` / ``col`` are local source / destination indices into ``node``; ``edge`` are positions in
the original ``edge_index``.  Uniform without replacement, directed, non-disjoint.

Parity status: UNPINNED against the reference sampler — ``pyg-lib`` / ``torch-sparse`` are not
installable in the build container, so ``torch.ops.pyg.neighbor_sample`` cannot be run; its RNG
stream is implementation-defined anyway.  The tests pin the *contract* instead: every sampled edge
exists, per-destination counts equal ``min(deg, k)`` without duplicates, hop structure, exact
k-hop equivalence for ``k = -1``, marginal uniformity (tests/test_gpu_sampler.py).
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor

from . import _native
from .edge_index import as_edge_index


@dataclass
class SamplerOutput:
    node: Tensor
    row: Tensor
    col: Tensor
    edge: Optional[Tensor]
    batch: Optional[Tensor] = None
    num_sampled_nodes: Optional[List[int]] = None
    num_sampled_edges: Optional[List[int]] = None


class NeighborSampler:
    r"""k-hop uniform neighbour sampler on the GPU.

    Args:
        edge_index: ``[2, E]`` device tensor or :class:`EdgeIndex` handle (its destination-sorted
            form is the CSC ``colptr`` / ``row`` the reference's sampler consumes,
            sampler/utils.py:46-111).
        num_nodes: number of nodes of the (homogeneous) graph.
        num_neighbors: fan-out per hop; ``-1`` takes every in-neighbour.
        seed: base of the counter-based RNG; batch ``b`` uses ``seed + b``.
    """

    def __init__(self, edge_index, num_nodes: int, num_neighbors: List[int], seed: int = 0):
        graph = as_edge_index(edge_index, num_nodes, num_nodes)
        csc = graph.by_dst()
        self.graph = graph
        self.colptr, self.row, self.perm = csc.ptr, csc.idx, csc.perm
        self.num_nodes = num_nodes
        self.num_neighbors = list(num_neighbors)
        if any(k > _native._lib.load().pygamd_sample_max_fanout() for k in self.num_neighbors):
            raise ValueError('bounded fan-outs above 64 are not supported (use -1 for all)')
        self.seed = seed
        self._calls = 0
        # global -> local id map; the dtype's minimum = not in the current batch (it must sort
        # below every claim value of pygamd_relabel); reset after every batch
        self._unset = torch.iinfo(self.colptr.dtype).min
        self._local = torch.full((num_nodes, ), self._unset, dtype=self.colptr.dtype,
                                 device=self.colptr.device)

    @torch.no_grad()
    def sample_from_nodes(self, seeds: Tensor, seed: Optional[int] = None) -> SamplerOutput:
        dev, dt = self.colptr.device, self.colptr.dtype
        seeds = seeds.to(device=dev, dtype=dt).contiguous()
        rng = self.seed + self._calls if seed is None else seed
        self._calls += 1
        local = self._local
        n_nodes = seeds.numel()
        local[seeds] = torch.arange(n_nodes, dtype=dt, device=dev)
        nodes, rows, cols, edges = [seeds], [], [], []
        num_nodes_hop, num_edges_hop = [n_nodes], []
        frontier, frontier_base = seeds, 0
        for hop, k in enumerate(self.num_neighbors):
            if frontier.numel() == 0:
                num_nodes_hop.append(0)
                num_edges_hop.append(0)
                continue
            cnt = _native.sample_counts(self.colptr, frontier, k)
            offsets = torch.zeros(frontier.numel() + 1, dtype=dt, device=dev)
            torch.cumsum(cnt, 0, out=offsets[1:])
            total = int(offsets[-1])  # host sync: sizes the hop's outputs (the reference's
            #                           CPU sampler is synchronous at the same point)
            src_g, dstpos, slot = _native.sample_neighbors(
                self.colptr, self.row, frontier, offsets, total, max(k, 0),
                (rng * 1_000_003 + hop) & 0x7FFFFFFFFFFFFFFF)
            # relabel: new nodes = sampled sources not seen yet, in order of first appearance
            new, row_local = _native.relabel_new_nodes(src_g, local, n_nodes)
            rows.append(row_local)
            cols.append(dstpos + frontier_base)
            edges.append(self.perm[slot])
            nodes.append(new)
            num_nodes_hop.append(new.numel())
            num_edges_hop.append(total)
            frontier, frontier_base = new, n_nodes
            n_nodes += new.numel()
        node = torch.cat(nodes)
        local[node] = self._unset  # leave the map clean for the next batch
        cat = (lambda xs: torch.cat(xs) if xs else torch.empty(0, dtype=dt, device=dev))
        return SamplerOutput(node=node, row=cat(rows), col=cat(cols), edge=cat(edges),
                             num_sampled_nodes=num_nodes_hop, num_sampled_edges=num_edges_hop)

"""Mini-batch loader over a device-resident graph: the ``NeighborLoader`` role
(torch_geometric/loader/neighbor_loader.py, node_loader.py:90-207, loader/utils.py:32-83,159) for
BASELINE config 4 — seeds are drawn per batch, the k-hop neighbourhood is sampled ON THE GPU
(:mod:`.sampler`), features are gathered with the HIP gather kernel (``filter_data``'s
``x[n_id]``) and the batch never touches the host."""
import queue
import threading
from dataclasses import dataclass
from typing import Iterator, List, Optional

import torch
from torch import Tensor

import os

from . import _native
from .edge_index import EdgeIndex
from .sampler import NeighborSampler

# collate_slots: layer 0 of the slot stack gathers its neighbours' rows straight from the feature
# matrix (only the destination rows of a batch are copied) — slots.SlotSampler.gather(direct=True).
# PYGAMD_SLOTS_DIRECT=0 copies every row of the batch first, as rounds 4-5 did.
SLOTS_DIRECT = os.environ.get('PYGAMD_SLOTS_DIRECT', '1') != '0'


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
    num_sampled_nodes: Optional[List[int]]
    num_sampled_edges: Optional[List[int]]
    batch: Optional[Tensor] = None  # disjoint sampling: the seed (tree) index of every node

    def record_stream(self, stream) -> None:
        for t in (self.x, self.y, self.edge_index, self.n_id, self.e_id, self.input_id,
                  self.batch):
            if isinstance(t, Tensor) and t.is_cuda:
                t.record_stream(stream)
        self.graph.record_stream(stream)


@dataclass
class PaddedBatch:
    """A mini-batch at STATIC shapes (``NeighborLoader.collate_padded``): ``hops`` is the
    sampler's padded-id output (block positions as node ids, per-hop CSR pointers, device-side
    counts), ``x`` the features of all ``hops.bases[-1]`` padded rows (padding rows repeat node 0),
    ``y`` the labels of the seeds.  Nothing in it depends on a host read: sampling, gather, the
    padded hop stack (nn/models/_fused_sage_hops.py) and the optimizer step replay as one
    hipGraph."""
    x: Tensor
    y: Optional[Tensor]
    hops: object
    n_id: Tensor
    batch_size: int


class NeighborLoader:
    r"""Iterates over mini-batches of ``batch_size`` seed nodes with their sampled ``k``-hop
    neighbourhoods.

    Args:
        x, y: node features ``[N, F]`` (fp32, device) and optional labels ``[N]``.
        edge_index: ``[2, E]`` device tensor.
        num_neighbors: fan-out per hop (``-1`` = all).
        input_nodes: seed pool (default: all nodes); shard it across ranks with
            :func:`pytorch_geometric_amd.data_parallel.shard_seeds`.
        prefetch: number of batches sampled AHEAD of the consumer (0 = sample inside
            ``__next__``).  With ``prefetch > 0`` a producer thread samples and gathers on its own
            HIP stream while the caller trains on the previous batch — the role the reference
            gives to ``num_workers`` DataLoader processes (loader/node_loader.py:90-152), without
            leaving the device.  Batches are handed over with an event the consumer's stream
            waits on.
        replace, disjoint, subgraph_type: the sampler options of the reference's loader
            (loader/neighbor_loader.py:209-233; see :class:`~.sampler.NeighborSampler`).
    """

    def __init__(self, x: Tensor, edge_index: Tensor, num_neighbors: List[int],
                 batch_size: int = 1024, y: Optional[Tensor] = None,
                 input_nodes: Optional[Tensor] = None, shuffle: bool = False,
                 drop_last: bool = False, seed: int = 0, prefetch: int = 0,
                 replace: bool = False, disjoint: bool = False,
                 subgraph_type: str = 'directional'):
        self.prefetch = int(prefetch)
        self._side = None
        self._slots = None  # the static-shape sampler of `collate_slots`, built on first use
        self.x, self.y = x, y
        self.num_nodes = x.size(0)
        self.sampler = NeighborSampler(edge_index, self.num_nodes, num_neighbors, seed=seed,
                                       replace=replace, disjoint=disjoint,
                                       subgraph_type=subgraph_type)
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
        bounded = min(fan) >= 0 and self.sampler.subgraph_type == 'directional'
        graph = EdgeIndex.from_sorted_batch(
            ei, out.node.numel(), max_in_degree=max(fan) if bounded else None)
        return Batch(x=x, y=y, edge_index=ei, graph=graph, n_id=out.node,
                     e_id=out.edge, input_id=seeds if input_id is None else input_id,
                     batch_size=seeds.numel(), num_sampled_nodes=out.num_sampled_nodes,
                     num_sampled_edges=out.num_sampled_edges, batch=out.batch)

    def collate_padded(self, seeds: Tensor, seed: int = 0,
                       seed_dev: Optional[Tensor] = None) -> PaddedBatch:
        """One batch at its static capacity, without any host synchronisation (bounded fan-outs,
        directional, non-disjoint).  ``seed_dev`` (int64 [1], device) is added to the RNG seed on
        the device: a captured step bumps it before every replay."""
        p = self.sampler.sample_padded(seeds, seed=seed, padded_ids=True, seed_dev=seed_dev,
                                       want_edge_ids=False)
        n_id = torch.cat([p.seeds] + p.new_nodes)
        x = _native.gather_rows(self.x, n_id)  # filter_data: x[n_id]
        y = None if self.y is None else self.y[seeds]
        return PaddedBatch(x=x, y=y, hops=p, n_id=n_id, batch_size=seeds.numel())

    def collate_slots(self, seeds: Tensor, epoch_dev: Tensor, with_labels: bool = True):
        """One batch in the static-shape SLOT layout (:mod:`pytorch_geometric_amd.slots`; bounded
        fan-outs, directional, non-disjoint, without replacement): 1 + 2 launches per hop for the
        sampling, 3 for the transposed CSRs of the backward, 1 for the feature gather, no host
        synchronisation.  ``epoch_dev``: int64 [1] on the device, >= 1, growing from batch to batch
        (a captured step bumps it before every replay).  Returns a ``SlotBatch`` with ``x`` = the
        gathered ``[R, 2 F]`` buffer and ``y`` = the seeds' labels (``with_labels=False``: none
        — ``slots.SlotTrainer`` reads them from ``self.y`` inside its loss launch)."""
        from .slots import SlotPlan, SlotSampler
        smp = self.sampler
        if smp.replace or smp.disjoint or smp.subgraph_type != 'directional' \
                or any(k < 1 for k in smp.num_neighbors):
            raise ValueError("'collate_slots' covers bounded fan-outs, directional, non-disjoint, "
                             "without replacement")
        if self._slots is None or self._slots.plan.B != seeds.numel():
            plan = SlotPlan(seeds.numel(), smp.num_neighbors, self.x.device)
            self._slots = SlotSampler(smp.colptr, smp.row, self.num_nodes, plan, seed=smp.seed)
        b = self._slots.sample(seeds, epoch_dev)
        b.x = self._slots.gather(self.x, b, direct=SLOTS_DIRECT and len(smp.num_neighbors) > 0)
        b.y = None if (self.y is None or not with_labels) else self.y[seeds]
        return b

    def _plan(self):
        n = self.input_nodes.numel()
        order = (torch.randperm(n, generator=self._gen).to(self.input_nodes.device)
                 if self.shuffle else torch.arange(n, device=self.input_nodes.device))
        nodes = self.input_nodes[order]   # ONE index launch per epoch; a batch's seeds are a view
        for b in range(len(self)):
            lo, hi = b * self.batch_size, (b + 1) * self.batch_size
            yield nodes[lo:hi], order[lo:hi]

    def __iter__(self) -> Iterator[Batch]:
        if self.prefetch <= 0:
            for seeds, sel in self._plan():
                yield self.collate(seeds, sel)
            return
        yield from self._prefetching_iter()

    def _prefetching_iter(self) -> Iterator[Batch]:
        dev = self.x.device
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        side = self._side
        plan = self._plan()
        first = next(plan, None)   # the seed tensors are made on the consumer's stream ...
        side.wait_stream(torch.cuda.current_stream(dev))  # ... before the producer reads them
        ready: 'queue.Queue' = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                torch.cuda.set_device(dev)
                item = first
                with torch.cuda.stream(side):
                    while item is not None and not stop.is_set():
                        batch = self.collate(*item)
                        done = torch.cuda.Event()
                        done.record(side)
                        if not put((batch, done)):
                            return
                        item = next(plan, None)
                put(None)
            except BaseException as exc:  # surfaced in the consumer
                put(exc)

        worker = threading.Thread(target=produce, name='pyg-amd-sampler', daemon=True)
        worker.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, done = item
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(done)
                batch.record_stream(cur)
                yield batch
        finally:
            stop.set()
            worker.join(timeout=10.0)

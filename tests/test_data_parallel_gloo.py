"""world_size-2 data parallelism on CPU (gloo): seed sharding + per-rank sampled subgraphs + ONE
flat-bucket all-reduce must reproduce the single-process GraphSAGE step on the whole graph (DDP
semantics: average over ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


N_NODES, N_EDGES, N_FEAT, N_CLASSES, N_SEEDS = 600, 1500, 6, 4, 40


def _graph():
    g = torch.Generator().manual_seed(0)
    ei = torch.randint(0, N_NODES, (2, N_EDGES), generator=g)
    x = torch.randn(N_NODES, N_FEAT, generator=g)
    y = torch.randint(0, N_CLASSES, (N_NODES, ), generator=g)
    seeds = torch.randperm(N_NODES, generator=g)[:N_SEEDS]
    return ei, x, y, seeds


class Sage2(torch.nn.Module):
    """2-layer GraphSAGE (mean aggregation, ReLU) on CPU tensors: the layers are the oracle's
    restatement of SAGEConv (this package's own layers compute on the HIP device only)."""

    def __init__(self):
        super().__init__()
        dims = [N_FEAT, 16, N_CLASSES]
        self.lin_l = torch.nn.ModuleList(torch.nn.Linear(a, b) for a, b in zip(dims, dims[1:]))
        self.lin_r = torch.nn.ModuleList(torch.nn.Linear(a, b, bias=False)
                                         for a, b in zip(dims, dims[1:]))

    def forward(self, x, edge_index):
        from oracle import pyg_oracle as O
        for i, (l, r) in enumerate(zip(self.lin_l, self.lin_r)):
            x = O.sage_conv(x, edge_index, l.weight, l.bias, r.weight)
            if i == 0:
                x = x.relu()
        return x


def _make_model():
    torch.manual_seed(3)
    return Sage2()


def _two_hop_batch(ei, seeds):
    """What NeighborLoader([-1, -1]) hands a rank for its seeds (loader/neighbor_loader.py: full
    fan-out): the seeds first, then the nodes two hops of in-neighbours reach, with every edge
    that points INTO the seeds or their in-neighbours, relabelled — the computation subgraph of a
    2-layer model on these seeds."""
    src, dst = ei
    in_batch = torch.zeros(N_NODES, dtype=torch.bool)
    in_batch[seeds] = True
    order = [seeds]
    frontier = seeds
    keep = torch.zeros(ei.size(1), dtype=torch.bool)
    for _ in range(2):
        is_dst = torch.zeros(N_NODES, dtype=torch.bool)
        is_dst[frontier] = True
        e = is_dst[dst]
        keep |= e
        new = torch.unique(src[e])
        new = new[~in_batch[new]]
        in_batch[new] = True
        order.append(new)
        frontier = torch.cat([frontier, new])
    n_id = torch.cat(order)
    local = torch.full((N_NODES, ), -1, dtype=torch.long)
    local[n_id] = torch.arange(n_id.numel())
    return n_id, local[ei[:, keep]]


def _worker(rank, world, port, queue):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pytorch_geometric_amd.data_parallel import (FlatGradBucket, broadcast_parameters,
                                                     shard_seeds)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix that
    model = Sage2()
    if rank == 0:
        model.load_state_dict(_make_model().state_dict())
    broadcast_parameters(model)
    bucket = FlatGradBucket(model)
    ei, X, Y, all_seeds = _graph()
    seeds = shard_seeds(all_seeds, rank, world)
    n_id, sub_ei = _two_hop_batch(ei, seeds)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(2):
        bucket.zero_()
        out = model(X[n_id], sub_ei)[:seeds.numel()]
        loss = torch.nn.functional.cross_entropy(out, Y[seeds])
        loss.backward()
        assert bucket.check_views()
        bucket.all_reduce_mean()
        opt.step()
    # plain lists: tensor hand-over through mp queues needs fd passing, unavailable in sandboxes
    queue.put((rank, seeds.tolist(), n_id.numel(), bucket.flat.tolist(),
               torch.cat([p.detach().reshape(-1) for p in model.parameters()]).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_seed_sharded_graphsage_matches_single_process():
    """The N = 2 path of BASELINE config 4 on CPU: every rank trains a 2-layer GraphSAGE on the
    sampled computation subgraph of ITS shard of the seeds; ONE flat-bucket all-reduce per step
    (DDP's average) must reproduce the single-process step on the whole graph and all seeds."""
    world = 2
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([queue.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    results = [(r, s, n, torch.tensor(gr), torch.tensor(w)) for r, s, n, gr, w in results]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, X, Y, all_seeds = _graph()
    half = N_SEEDS // 2
    assert results[0][1] == all_seeds[:half].tolist() and results[1][1] == all_seeds[half:].tolist()
    # (the batches are real subgraphs: each rank touched a part of the graph, not all of it)
    assert all(half < r[2] < N_NODES for r in results)
    # both ranks hold identical gradients and weights after the collective
    assert torch.allclose(results[0][3], results[1][3], atol=1e-7)
    assert torch.allclose(results[0][4], results[1][4], atol=1e-7)
    # single-process reference: the mean over ranks of the per-shard mean losses == the mean over
    # all seeds (equal shards), evaluated full-batch on the whole graph
    model = _make_model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(X, ei)[all_seeds], Y[all_seeds])
        loss.backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.allclose(results[0][4], ref, atol=1e-6)


def test_shard_seeds_covers_everything_once():
    from pytorch_geometric_amd.data_parallel import shard_seeds
    idx = torch.arange(1027)
    for world in (1, 2, 3, 8):
        parts = [shard_seeds(idx, r, world) for r in range(world)]
        assert torch.equal(torch.cat(parts), idx)
        assert max(p.numel() for p in parts) - min(p.numel() for p in parts) <= world

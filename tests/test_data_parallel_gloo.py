"""world_size-2 data parallelism on CPU (gloo): seed sharding + ONE flat-bucket all-reduce must
reproduce single-process full-batch gradients (DDP semantics: average over ranks)."""
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


def _make_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))


def _worker(rank, world, port, queue):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pytorch_geometric_amd.data_parallel import (FlatGradBucket, broadcast_parameters,
                                                     shard_seeds)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix that
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    if rank == 0:
        model.load_state_dict(_make_model().state_dict())
    broadcast_parameters(model)
    bucket = FlatGradBucket(model)
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(40, 6, generator=g), torch.randint(0, 4, (40, ), generator=g)
    seeds = shard_seeds(torch.arange(40), rank, world)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(2):
        bucket.zero_()
        loss = torch.nn.functional.cross_entropy(model(X[seeds]), Y[seeds])
        loss.backward()
        assert bucket.check_views()
        bucket.all_reduce_mean()
        opt.step()
    # plain lists: tensor hand-over through mp queues needs fd passing, unavailable in sandboxes
    queue.put((rank, seeds.tolist(), bucket.flat.tolist(),
               torch.cat([p.detach().reshape(-1) for p in model.parameters()]).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_flat_bucket_matches_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([queue.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    results = [(r, s, torch.tensor(gr), torch.tensor(w)) for r, s, gr, w in results]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == list(range(20)) and results[1][1] == list(range(20, 40))
    # both ranks hold identical gradients and weights after the collective
    assert torch.allclose(results[0][2], results[1][2], atol=1e-7)
    assert torch.allclose(results[0][3], results[1][3], atol=1e-7)
    # single-process reference: mean over ranks of per-shard mean losses == full-batch mean here
    model = _make_model()
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(40, 6, generator=g), torch.randint(0, 4, (40, ), generator=g)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(X), Y)
        loss.backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.allclose(results[0][3], ref, atol=1e-6)


def test_shard_seeds_covers_everything_once():
    from pytorch_geometric_amd.data_parallel import shard_seeds
    idx = torch.arange(1027)
    for world in (1, 2, 3, 8):
        parts = [shard_seeds(idx, r, world) for r in range(world)]
        assert torch.equal(torch.cat(parts), idx)
        assert max(p.numel() for p in parts) - min(p.numel() for p in parts) <= world

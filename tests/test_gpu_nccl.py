"""The RCCL ('nccl' backend) code path on ONE GPU (the only multi-rank runs so far
were gloo on the CPU): a process group of one rank over RCCL, the parameter broadcast and the
flat-bucket gradient all-reduce on device tensors, and one bench.py step in both modes launched the
way the driver launches N ranks (torch.distributed.run, rendezvous on 127.0.0.1).  No scaling
claim — a world of one rank moves no data between GPUs; what is covered is that the collectives are
issued on device buffers through RCCL and that the bench line carries the per-rank timings."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from pytorch_geometric_amd.data_parallel import FlatGradBucket, broadcast_parameters, shard_seeds
from pytorch_geometric_amd.nn import GraphSAGE
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = GraphSAGE(16, 32, num_layers=2, out_channels=5).to(dev)
before = [p.detach().clone() for p in model.parameters()]
broadcast_parameters(model)                       # dist.broadcast on a device buffer
assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
bucket = FlatGradBucket(model)
bucket.flat.copy_(torch.arange(bucket.flat.numel(), device=dev, dtype=torch.float32))
want = bucket.flat.clone()
work = bucket.all_reduce_mean(force=True)         # dist.all_reduce(SUM) / world on the flat bucket
assert work is None or work.is_completed() or True
torch.cuda.synchronize()
assert torch.equal(bucket.flat, want) and bucket.check_views()
h = bucket.all_reduce_mean(async_op=True, force=True)
h.wait()
assert torch.equal(bucket.flat, want)
t = torch.ones(3, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
assert shard_seeds(torch.arange(10), 0, 1).numel() == 10
dist.destroy_process_group()
print('NCCL_WORLD1_OK')
'''


_CAPTURE_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from pytorch_geometric_amd.data_parallel import FlatGradBucket
from pytorch_geometric_amd.hipgraph import CapturedStep
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(512, 32, generator=g).to(dev)
y = torch.randn(512, 8, generator=g).to(dev)

def make():
    torch.manual_seed(1)
    m = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).to(dev)
    b = FlatGradBucket(m)
    o = torch.optim.Adam(m.parameters(), lr=1e-2, capturable=True, fused=True)
    def step():
        b.zero_()
        torch.nn.functional.mse_loss(m(x), y).backward()
        b.all_reduce_mean(force=True)      # RCCL all-reduce of the flat bucket, stream-ordered
        o.step()
    return m, b, step

m_eager, _, step_eager = make()
for _ in range(6):
    step_eager()
m_cap, bucket, step_cap = make()
replay = CapturedStep(step_cap, warmup=3)   # 3 eager steps, then ONE graph: fwd + bwd + all-reduce + Adam
for _ in range(3):
    replay()
torch.cuda.synchronize()
assert bucket.check_views()
for a, b in zip(m_eager.parameters(), m_cap.parameters()):
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()
dist.barrier()
dist.destroy_process_group()
print('NCCL_CAPTURE_OK')
'''


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _env():
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        e.pop(k, None)
    e['MASTER_ADDR'] = '127.0.0.1'
    e['MASTER_PORT'] = str(_free_port())
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return e


@pytest.mark.timeout(300)
def test_rccl_collectives_on_one_rank():
    res = subprocess.run([sys.executable, '-c', _WORKER, ROOT], capture_output=True, text=True,
                         env=_env(), timeout=280, cwd=ROOT)
    assert res.returncode == 0 and 'NCCL_WORLD1_OK' in res.stdout, res.stderr[-3000:]


@pytest.mark.timeout(300)
def test_rccl_all_reduce_and_adam_capture_into_one_hipgraph():
    """What the 8-rank mini-batch step replays: forward + backward + the RCCL
    all-reduce of the flat gradient bucket + the fused Adam update recorded into ONE hipGraph
    (RCCL launches are stream-ordered kernels: they capture like any other), replayed, and equal
    to the same steps issued eagerly."""
    res = subprocess.run([sys.executable, '-c', _CAPTURE_WORKER, ROOT], capture_output=True,
                         text=True, env=_env(), timeout=280, cwd=ROOT)
    assert res.returncode == 0 and 'NCCL_CAPTURE_OK' in res.stdout, res.stderr[-3000:]


@pytest.mark.timeout(600)
@pytest.mark.parametrize('mode', ['fullbatch', 'minibatch'])
def test_bench_step_under_torchrun_with_rccl(mode):
    e = _env()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', e['MASTER_PORT'],
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--init-dist', '--mode', mode,
           '--steps', '2', '--warmup', '1', '--scale', '0.02', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=560, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['steps'] == 2 and out['value'] > 0
    pr = out['config']['per_rank_ms_per_step']
    assert len(pr['ranks']) == 1 and pr['min'] == pr['max'] > 0
    assert out['config']['allreduce_ms_per_step'] > 0      # the collective really ran
    if mode == 'fullbatch':
        assert out['config']['process_group'] == 'nccl'


@pytest.mark.timeout(600)
def test_captured_minibatch_step_holds_the_collective_under_rccl():
    """`bench.py --mode minibatch --capture` with a process group over RCCL: sampling, gather,
    forward, backward, the gradient all-reduce and Adam replay as ONE hipGraph per batch — what
    every rank of an 8-GPU run executes."""
    e = _env()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', e['MASTER_PORT'],
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--init-dist', '--mode', 'minibatch',
           '--capture', '--steps', '4', '--warmup', '2', '--scale', '0.02', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=560, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['value'] > 0
    assert 'RCCL all-reduce + Adam = one hipGraph' in out['config']['captured'], out['config']


@pytest.mark.timeout(900)
def test_bench_two_ranks_share_the_one_gpu():
    """The N > 1 COMPUTE path on hardware: two ranks of `bench.py --gpus 2`, both on GPU 0
    (PYGAMD_BENCH_SHARE_GPU=1: collectives over gloo on device tensors — RCCL refuses two ranks on
    one device).  Not a measurement; what is covered is everything the 8-GPU run does around the
    kernels: per-rank graph replicas, parameter broadcast, the flat-bucket all-reduce every step,
    barrier + max-over-ranks timing, one JSON line from rank 0 with both ranks' timings, and the
    parity leg against the committed reference sample."""
    e = _env()
    e['PYGAMD_BENCH_SHARE_GPU'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', e['MASTER_PORT'],
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--scale', '0.02', '--no-cpu-baseline']
    res = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=860, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0
    pr = out['config']['per_rank_ms_per_step']
    assert len(pr['ranks']) == 2 and 0 < pr['min'] <= pr['max']
    assert abs(out['ms_per_step'] - pr['max']) < 1e-6 * max(pr['max'], 1.0) + 1e-3
    assert out['config']['allreduce_ms_per_step'] > 0
    assert out['config']['process_group'] == 'gloo'
    assert out['scaling'] == 'weak'

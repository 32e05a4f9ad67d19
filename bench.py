#!/usr/bin/env python
"""bench.py — aggregated edges/sec, forward+backward, 3-layer GraphSAGE on a synthetic
ogbn-products-shaped graph (BASELINE.json `metric`, `configs[1]`), 1..N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: re-executes itself under
                                                          torch.distributed.run with N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = zero grads, full-batch forward of GraphSAGE(100, 256, 3 layers, 47 classes),
cross-entropy on random labels of the 8 % training split, backward, (N > 1: ONE flat-bucket
gradient all-reduce over RCCL), Adam update.  Inputs are resident in HBM before the timed region.  N > 1 is plain data parallelism
over graph replicas: each rank owns its own synthetic graph of the same shape (seed + rank), so
per-GPU work is fixed ("weak").  value = L * E * N / t_step.

Prints ONE JSON line on rank 0, including `roofline` (dominant kernel = the F=256 CSR SpMM,
HIP-event timed inside the timed region) and `cpu_baseline` (the oracle's CPU scatter path on a
bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def spmm_algorithmic_bytes(info) -> float:
    """E*(4F + b) + (N_rows + 1)*b + N_rows*4F  (+ N*4 for the mean's degree
    vector when it is read as a per-source scale, + N_rows*4F when the launch accumulates onto
    its output, i.e. the fused `grad_root + A^T grad_agg` of the backward, + N_rows*4F when its
    epilogue reads a ReLU output — or, 32 x smaller, its one-bit-per-element mask — to apply
    that activation's backward)."""
    b, Fw = info['idx_bytes'], info['F']
    total = info['nnz'] * (4 * Fw + b) + (info['n_rows'] + 1) * b + info['n_rows'] * 4 * Fw
    if info['src_scale']:
        total += info['n_src'] * 4
    if info['weighted']:
        total += info['nnz'] * 4
    if info.get('accumulate'):
        total += info['n_rows'] * 4 * Fw  # out += result: the old rows are read as well
    if info.get('relu_mask'):
        total += info['n_rows'] * 4 * Fw  # the activation rows the fused ReLU backward reads
    if info.get('relu_bits'):
        total += info['n_rows'] * 4 * ((Fw + 31) // 32)  # ... or one bit per element
    return float(total)


def pmc_traffic(args, N, E, F_dom):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r02_pmc_spmm_f256.json: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, collected by
    scripts/gpu_r02_profile.sh over this very command).  Counters cannot be read from inside this
    process, so the number is only reported when this run is the workload the counters were
    collected on; otherwise null."""
    path = os.path.join(ROOT, 'profiles', 'r02_pmc_spmm_f256.json')
    try:
        with open(path) as f:
            d = json.load(f)
    except OSError:
        return None
    w = d['workload']
    same = (w['N'] == N and w['E'] == E and w['F'] == F_dom and not args.uniform
            and w['index_dtype'] == args.index_dtype)
    return d['traffic_bytes_per_launch'] if same else None


def _time_steps(step_fn, steps: int, warmup: int = 1):
    """Median (fwd, bwd) wall seconds of `steps` calls after `warmup`."""
    rec = []
    for it in range(warmup + steps):
        tf, tb = step_fn()
        if it >= warmup:
            rec.append((tf + tb, tf, tb))
    rec.sort()
    return rec[len(rec) // 2]


def cpu_baseline(scale: float, steps: int = 3):
    """PyG's CPU paths timed on this box's host cores, on a `scale` x products-shaped sample of the
    bench workload: the UNMODIFIED reference (oracle/_ref, staged by
    oracle/make_ref.py; kind "reference") — primary line = the scatter path (plain `edge_index`
    tensor: index_select + scatter_add_), second line = its fast path (`adj_t` as a torch CSR
    tensor -> torch.sparse.mm).  Falls back to the oracle port (kind "port") only when no staged
    reference travelled with the snapshot."""
    from pytorch_geometric_amd.datasets import products_like
    x, y, ei, c = products_like(seed=1, scale=scale)
    E = ei.size(1)
    g = torch.Generator().manual_seed(7)
    train_idx = torch.randperm(x.size(0), generator=g)[:max(int(0.0803 * x.size(0)), 1)]
    threads = torch.get_num_threads()
    base = {'unit': 'edges/s', 'cores': threads, 'host_cpus': os.cpu_count()}
    shape = f'{scale:g} x ogbn-products shape: N={x.size(0)}, E={E}'
    try:
        from oracle import make_ref
        make_ref.import_reference()
        from torch_geometric.nn import GraphSAGE as RefSAGE
        from torch_geometric.utils import to_torch_csc_tensor
    except ImportError:
        RefSAGE = None
    if RefSAGE is None:
        from oracle import pyg_oracle as O
        from pytorch_geometric_amd.nn import GraphSAGE
        torch.manual_seed(0)
        model = GraphSAGE(100, 256, num_layers=3, out_channels=c)
        params = [(cv.lin_l.weight, cv.lin_l.bias, cv.lin_r.weight) for cv in model.convs]

        def step():
            t0 = time.perf_counter()
            for p in model.parameters():
                p.grad = None
            loss = F.cross_entropy(O.graphsage(x, ei, params)[train_idx], y[train_idx])
            t1 = time.perf_counter()
            loss.backward()
            return t1 - t0, time.perf_counter() - t1

        t, tf, tb = _time_steps(step, steps)
        return dict(base, value=3 * E / t, kind='port',
                    sample=(f'oracle/pyg_oracle.py graphsage fwd+bwd (index_select + scatter_add_ '
                            f'mean), {shape}, median of {steps} steps after 1 warm-up, '
                            f'{t * 1e3:.0f} ms/step'))
    torch.manual_seed(0)
    model = RefSAGE(100, 256, num_layers=3, out_channels=c)

    def make_step(graph):
        def step():
            t0 = time.perf_counter()
            model.zero_grad(set_to_none=True)
            loss = F.cross_entropy(model(x, graph)[train_idx], y[train_idx])
            t1 = time.perf_counter()
            loss.backward()
            return t1 - t0, time.perf_counter() - t1
        return step

    t, tf, tb = _time_steps(make_step(ei), steps)
    adj_t = to_torch_csc_tensor(ei, size=(x.size(0), x.size(0))).t()  # CSR
    t2, tf2, tb2 = _time_steps(make_step(adj_t), steps)
    return dict(
        base, value=3 * E / t, kind='reference',
        sample=(f'unmodified torch_geometric 2.9.0 (oracle/_ref) GraphSAGE(100,256,3,{c}) fwd + '
                f'CE(8% split) + bwd, plain edge_index -> index_select + scatter_add_ (the CPU '
                f'scatter path), {shape}, median of {steps} steps after 1 warm-up: '
                f'fwd {tf:.2f} s + bwd {tb:.2f} s; torch threads {threads} of '
                f'{os.cpu_count()} host CPUs'),
        csr_fast_path={'value': 3 * E / t2, 'unit': 'edges/s',
                       'sample': f'same model and sample, adj_t = torch.sparse_csr -> '
                                 f'torch.sparse.mm: fwd {tf2:.2f} s + bwd {tb2:.2f} s'})


def run_minibatch(args, rank, local_rank, world, dev):
    """BASELINE config 4 (informational second mode; the default mode is the metric's config):
    GraphSAGE(128, 256, 3 layers, 172 classes) + NeighborLoader [15, 10, 5], batch 1024 per rank,
    on a synthetic papers100M-shaped graph replicated on every GPU; seeds are sharded across ranks
    (examples/multi_gpu/distributed_sampling.py:70-71), sampling and feature gather run on the
    GPU, the only exchange is ONE flat-bucket gradient all-reduce per step."""
    import torch.distributed as dist

    from pytorch_geometric_amd.data_parallel import (FlatGradBucket, broadcast_parameters,
                                                     shard_seeds)
    from pytorch_geometric_amd.datasets import powerlaw_undirected
    from pytorch_geometric_amd.loader import NeighborLoader
    from pytorch_geometric_amd.nn import GraphSAGE
    # the FULL ogbn-papers100M shape by default: x (56.9 GB) + edge_index (25.9 GB) + the CSC form
    # (26.7 GB) are resident in the 288 GB of one MI355X, replicated per rank;
    # the graph is generated ON the device (1.6 G edges are too many to draw on the host)
    scale = args.scale
    N = int(111_059_956 * scale)
    E = int(1_615_685_872 * scale) // 2 * 2
    t_gen = time.perf_counter()
    ei = powerlaw_undirected(N, E, seed=3, device=dev)  # same graph on every rank (replicated)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(N, 128, device=dev, generator=g)
    y = torch.randint(0, 172, (N, ), device=dev, generator=g)
    train_idx = torch.randperm(N, generator=torch.Generator().manual_seed(11))[:max(N // 90, 1024)]
    seeds = shard_seeds(train_idx, rank, world)
    fan = [15, 10, 5]
    loader = NeighborLoader(x, ei, fan, batch_size=1024, y=y, input_nodes=seeds, shuffle=True,
                            drop_last=True, seed=17 + rank, prefetch=args.prefetch)
    torch.cuda.synchronize(dev)
    t_gen = time.perf_counter() - t_gen
    torch.cuda.empty_cache()  # the radix sort's scratch
    torch.manual_seed(0)
    model = GraphSAGE(128, 256, num_layers=3, out_channels=172).to(dev)
    broadcast_parameters(model)
    bucket = FlatGradBucket(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def batches():
        while True:
            for b in loader:
                yield b

    it = batches()
    edges = 0
    ar_events = []

    def step():
        nonlocal edges
        b = next(it)
        bucket.zero_()
        out = model(b.x, b.graph, num_sampled_nodes_per_hop=b.num_sampled_nodes,
                    num_sampled_edges_per_hop=b.num_sampled_edges)[:b.batch_size]
        loss = F.cross_entropy(out, b.y[:b.batch_size])
        loss.backward()
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean()
            e1.record()
            ar_events.append((e0, e1))
        opt.step()
        ne = b.num_sampled_edges  # layer l aggregates the edges of hops 0 .. L-1-l
        edges += sum(sum(ne[:len(ne) - l]) for l in range(len(ne)))
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    edges = 0
    del ar_events[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(loss).item()
    it.close()  # stops the loader's prefetch thread
    t = torch.tensor([elapsed, float(edges)], dtype=torch.float64, device=dev)
    if world > 1:
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        t[0] = tm[0]
    elapsed, total_edges = float(t[0]), float(t[1])
    if rank == 0:
        print(json.dumps({
            'metric': 'edges/sec (fwd+bwd) 3-layer SAGE + NeighborLoader [15,10,5], '
                      'papers100M shape (BASELINE config 4, informational)',
            'value': total_edges / elapsed, 'unit': 'edges/s',
            'n_gpus': dist.get_world_size() if dist.is_initialized() else 1,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'GraphSAGE(128->256->256->172) mini-batch training, batch '
                                   f'1024 seeds/rank, fan-out {fan}, trim_to_layer, synthetic '
                                   f'papers100M shape x {scale:g} (N={N}, E={E}) replicated per '
                                   f'GPU, GPU sampler + gather, {args.prefetch} batch(es) prefetched on a side stream',
                       'parallelism': f'dp{world} (seed sharding, one flat-bucket '
                                      f'all-reduce/step)',
                       'allreduce_ms_per_step': round(
                           sum(a.elapsed_time(b) for a, b in ar_events)
                           / max(len(ar_events), 1), 4) if world > 1 else 0.0,
                       'graph_build_s': round(t_gen, 1),
                       'hbm_gb_allocated': round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                       'gemm': gemm_desc(False)}}), flush=True)


def gemm_desc(tuned: bool) -> str:
    from pytorch_geometric_amd.nn.models import _fused_sage
    if _fused_sage.GEMM_BACKEND == 'own':
        from pytorch_geometric_amd import get_gemm_mode
        if get_gemm_mode() == 'split':
            return ('pytorch_geometric_amd/csrc/gemm.hip, PYGAMD_GEMM_MODE=split (NOT the '
                    'default): '
                    'fp32 operands as 3 bf16 terms each, 6 v_mfma_f32_32x32x16_bf16 products, '
                    'fp32 accumulation, in the stand-alone forward / dgrad / wgrad kernels; the '
                    'one-kernel layer forward stays on the fp32 instruction')
        return ('pytorch_geometric_amd/csrc/gemm.hip: hand-written fp32 MFMA '
                '(v_mfma_f32_32x32x2_f32) forward (+bias+ReLU epilogue), dgrad (+1/deg row '
                'scale and ReLU-backward epilogues) and split-reduction wgrad (+bias gradient) '
                'kernels')
    return ('rocBLAS/hipBLASLt via torch.mm (PYGAMD_GEMM=lib), ' +
            ('solution per shape from pytorch_geometric_amd/tuning (TunableOp, read-only)'
             if tuned else 'default heuristics'))


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks of this
    script under torch.distributed.run (one process per GPU) and pass everything through; rank 0
    of the child job prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def init_ranks(args):
    """(rank, local_rank, world, device).  Backend 'nccl' (= RCCL) on GPUs; 'gloo' only for the
    CPU launcher check (--dry-run), which runs no kernel of this repo."""
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    has_gpu = torch.cuda.is_available()
    if not has_gpu and not args.dry_run:
        raise RuntimeError('bench.py needs a GPU (there is no CPU fallback); --dry-run only '
                           'checks the multi-rank launcher')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if has_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl' if has_gpu else 'gloo', rank=rank, world_size=world)
        assert dist.get_world_size() == world
    if world != args.gpus:
        raise RuntimeError(f'--gpus {args.gpus} but the process group has {world} rank(s): '
                           f'launch with --nproc-per-node {args.gpus} (or let bench.py spawn the '
                           f'ranks itself by not setting WORLD_SIZE)')
    dev = torch.device('cuda', local_rank) if has_gpu else torch.device('cpu')
    if has_gpu:
        torch.cuda.set_device(dev)
    return rank, local_rank, world, dev


def run_dry(args, rank, world, dev):
    """Launcher / collective plumbing only (used by tests/test_bench_launcher.py on CPU with gloo
    and usable on GPUs): rendezvous, parameter broadcast, K flat-bucket all-reduces of a toy
    module, barrier + max-over-ranks timing, ONE JSON line on rank 0.  No kernel of this repo
    runs and nothing here is a measurement of the hot path."""
    import torch.distributed as dist

    from pytorch_geometric_amd.data_parallel import FlatGradBucket, broadcast_parameters
    torch.manual_seed(100 + rank)
    model = torch.nn.Linear(64, 32).to(dev)
    broadcast_parameters(model)
    bucket = FlatGradBucket(model)

    def fence():
        if world > 1:
            dist.barrier()
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        bucket.flat.fill_(float(rank + 1))
        bucket.all_reduce_mean()
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    expect = sum(range(1, world + 1)) / world
    ok = bool(torch.allclose(bucket.flat, torch.full_like(bucket.flat, expect)))
    w0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    same = w0.clone()
    if world > 1:
        dist.broadcast(same, src=0)
    ok = ok and bool(torch.equal(same, w0))
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        n_pg = dist.get_world_size() if dist.is_initialized() else 1
        print(json.dumps({
            'metric': 'launcher dry run (no kernels)', 'value': None, 'unit': None,
            'n_gpus': n_pg, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': float(t.item()) / max(args.steps, 1) * 1e3, 'dry_run': True,
            'backend': dist.get_backend() if dist.is_initialized() else None,
            'collectives_ok': bool(flag.item() == 1.0)}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if flag.item() != 1.0:
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=['fullbatch', 'minibatch'], default='fullbatch',
                    help="'fullbatch' = BASELINE config 2 (the metric's configuration); "
                         "'minibatch' = config 4, informational")
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--dry-run', action='store_true',
                    help='only exercise the N-rank launcher and the collectives (no kernels)')
    ap.add_argument('--scale', type=float, default=1.0, help='fraction of the products shape')
    ap.add_argument('--prefetch', type=int, default=2,
                    help='mini-batch mode: batches sampled ahead on a side stream (0 = inline)')
    ap.add_argument('--index-dtype', choices=['int64', 'int32'], default='int64')
    ap.add_argument('--uniform', action='store_true', help='uniform instead of power-law graph')
    ap.add_argument('--cpu-scale', type=float, default=1 / 16,
                    help='fraction of the products shape the CPU baseline is timed on '
                         '(BASELINE.md 3.5: s in {1 .. 1/16})')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-tuned-gemm', action='store_true',
                    help='use the default rocBLAS/hipBLASLt heuristics instead of the shipped '
                         'TunableOp table')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch.distributed as dist

    rank, local_rank, world, dev = init_ranks(args)
    if args.dry_run:
        return run_dry(args, rank, world, dev)

    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.data_parallel import FlatGradBucket, broadcast_parameters
    from pytorch_geometric_amd.datasets import products_like
    from pytorch_geometric_amd.nn import GraphSAGE

    if world > 1:  # the ranks build their synthetic graphs on the host at the same time
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    pga.load_library()  # fail loudly if the HIP library is missing
    tuned = False
    if not args.no_tuned_gemm:
        from pytorch_geometric_amd.tuning import enable_tuned_gemms
        tuned = enable_tuned_gemms()

    if args.mode == 'minibatch':
        run_minibatch(args, rank, local_rank, world, dev)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    idx_dtype = torch.int64 if args.index_dtype == 'int64' else torch.int32
    t_gen = time.perf_counter()
    x, y, ei, num_classes = products_like(seed=1 + rank, scale=args.scale,
                                          skewed=not args.uniform, dtype=idx_dtype)
    N, E = x.size(0), ei.size(1)
    x, y, ei = x.to(dev), y.to(dev), ei.to(dev)
    t_gen = time.perf_counter() - t_gen

    torch.manual_seed(0)
    model = GraphSAGE(100, 256, num_layers=3, out_channels=num_classes).to(dev)
    broadcast_parameters(model)
    bucket = FlatGradBucket(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    # like every full-batch PyG example, the loss is taken on the training split only
    # (ogbn-products: 196,615 of 2,449,029 nodes = 8 %)
    g = torch.Generator().manual_seed(7 + rank)
    train_idx = torch.randperm(N, generator=g)[:max(int(0.0803 * N), 1)].to(dev)
    y_train = y[train_idx]

    ar_events = []  # (start, end) HIP events around the gradient all-reduce, timed steps only

    def step():
        bucket.zero_()
        out = model(x, ei)
        loss = F.cross_entropy(out[train_idx], y_train)
        loss.backward()
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean()  # the current stream waits for RCCL's stream before e1
            e1.record()
            ar_events.append((e0, e1))
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    del ar_events[:]
    sink = []
    _native.timing_sink = sink
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    _native.timing_sink = None
    assert bucket.check_views(), 'gradient views were replaced: the all-reduce saw stale data'
    assert torch.isfinite(loss).item(), 'loss is not finite'

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = 3.0 * E * world / (elapsed / args.steps)

    allreduce_ms = (sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1)
                    if world > 1 else 0.0)
    n_pg = dist.get_world_size() if dist.is_initialized() else 1

    # ---- roofline of the dominant kernel: the CSR SpMM at F = 256 (2 of the 5 SpMM launches per
    # step — layer 2 forward and its transposed backward; layer 3 is re-ordered to width 48) ----
    groups, fused = {}, {}
    for info, ev0, ev1 in sink:
        dst = fused if info.get('fused_gemm') else groups
        dst.setdefault(info['F'], []).append((info, ev0.elapsed_time(ev1)))
    dom_F = 256 if 256 in groups else max(groups)
    dom = groups[dom_F]
    avg_ms = sum(ms for _, ms in dom) / len(dom)
    alg_bytes = sum(spmm_algorithmic_bytes(i) for i, _ in dom) / len(dom)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    spmm_ms_per_step = sum(ms for g in groups.values() for _, ms in g) / args.steps
    roofline = {
        'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(args, N, E, dom_F),
        'traffic_source': 'profiles/ PMC passes of this workload (FETCH_SIZE x 2 + WRITE_SIZE per '
                          'launch, collected by scripts/gpu_r02_profile.sh), not a live counter',
        'kernel': f'spmm_sum_rows<F={dom_F}> (pygamd_spmm_csr: the stand-alone CSR aggregation; '
                  f'with the layer forward fused into one kernel these launches are the '
                  f'transposed, accumulating backward with the ReLU-backward epilogue)',
        'launches_timed': len(dom), 'avg_launch_ms': round(avg_ms, 4),
        'algorithmic_bytes_per_launch': alg_bytes,
        'all_spmm_ms_per_step': round(spmm_ms_per_step, 3),
        'per_width_avg_ms': {str(k): round(sum(ms for _, ms in v) / len(v), 4)
                             for k, v in sorted(groups.items())},
    }
    if fused:
        # the one-kernel layer forward (csrc/sage_fused.hip) is bound by BOTH rooflines at once:
        # its gather phase by HBM, its transform phase by the fp32 matrix cores
        fl = {}
        for Fw, items in sorted(fused.items()):
            ms = sum(t for _, t in items) / len(items)
            i0 = items[0][0]
            Fo, K = i0['fused_gemm']['Fo'], i0['fused_gemm']['K']
            # gather + rowptr/col + root rows read + aggregated rows written once + output written
            byts = (spmm_algorithmic_bytes(i0) + i0['n_rows'] * 4 * Fw + i0['n_rows'] * 4 * Fo)
            flops = 2.0 * i0['n_rows'] * K * Fo
            fl[str(Fw)] = {'avg_launch_ms': round(ms, 4), 'launches_timed': len(items),
                           'hbm_algorithmic_bytes': byts,
                           'hbm_GBps': round(byts / (ms * 1e-3) / 1e9, 1),
                           'hbm_frac': round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           'mfma_TFLOPs': round(flops / (ms * 1e-3) / 1e12, 1),
                           'mfma_frac_of_157.3': round(flops / (ms * 1e-3) / 1e12 / 157.3, 4)}
        roofline['fused_layer_forward'] = {
            'kernel': 'sage_fused_fwd_kernel (pygamd_sage_layer_forward: aggregation + '
                      '[agg|x] @ W^T + bias + ReLU)', 'per_width': fl}

    if rank == 0:
        result = {
            'metric': 'edges/sec (fwd+bwd) 3-layer SAGE, ogbn-products shape',
            'value': value, 'unit': 'edges/s', 'n_gpus': n_pg, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': (f'GraphSAGE(100->256->256->{num_classes}, mean aggr) full-batch '
                             f'fwd+bwd(CE on an 8% train split)+Adam on a synthetic ogbn-products-shaped graph per GPU '
                             f'(N={N}, E={E}, {"uniform" if args.uniform else "power-law"} '
                             f'degrees, {args.index_dtype} edge_index, fp32 features)'),
                'edges_per_step_per_gpu': 3 * E, 'scale': args.scale,
                'parallelism': f'dp{world} (graph replicas, one flat-bucket all-reduce/step)',
                'allreduce_ms_per_step': round(allreduce_ms, 4),
                'graph_gen_s': round(t_gen, 1),
                'gemm': gemm_desc(tuned),
                'schedule': 'fused stack: [agg|x] single GEMM per layer; 256->47 layer '
                            'transforms first and aggregates at width 48; layers 1-2 forward '
                            'as ONE kernel each (aggregation -> LDS -> MFMA); ReLU backward in '
                            'the epilogue of the kernel producing each activation gradient '
                            '(transposed SpMM / dgrad GEMM); bias gradients from the '
                            'weight-gradient GEMM\'s pass over grad_out; the mean\'s 1/deg of '
                            'the backward applied in the dgrad GEMM epilogue',
            },
            'roofline': roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            result['cpu_baseline'] = cpu_baseline(args.cpu_scale)
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

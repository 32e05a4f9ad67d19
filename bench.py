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

Prints ONE JSON line on rank 0, including `roofline` (the kernel with the largest share of the
timed region — found from HIP events around every launch of this repo's kernels — against the
HBM roofline, plus whole-step aggregation / GEMM figures), `cpu_baseline` (the unmodified reference
on the host cores, on a bounded sample of the same workload) and `parity_at_cpu_scale` (the same
sample through the GPU stack: loss, output and every parameter gradient against the reference).
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
    # (src_bits: only the source rows with a non-zero entry are read — `live_nnz` stored entries
    # point at one — plus one bit per source row; every index still is)
    rows_read = info.get('live_nnz', info['nnz']) if info.get('src_bits') else info['nnz']
    total = (rows_read * 4 * Fw + info['nnz'] * b + (info['n_rows'] + 1) * b
             + info['n_rows'] * 4 * Fw)
    if info.get('src_bits'):
        total += info['n_src'] / 8
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


def pmc_traffic(args, N, E, kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this very command
    (profiles/r05_pmc_bench.json, r04_..., r03_...: FETCH_SIZE and WRITE_SIZE in separate passes, converted as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes; collected by scripts/gpu_r05_profile.sh).
    Counters cannot be read from inside this process, so a number is only reported when this run is
    the workload the counters were collected on; otherwise null."""
    if _live_traffic is not None:   # this run's own counter passes (live_pmc_traffic)
        return _live_traffic.get(kernel)
    d = None
    for name in ('r06_pmc_bench.json', 'r05_pmc_bench.json', 'r04_pmc_bench.json', 'r03_pmc_bench.json'):  # newest file that knows the kernel
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                cand = json.load(f)
        except (OSError, ValueError):
            continue
        if kernel in cand.get('traffic_bytes_per_launch', {}):
            d = cand
            break
    if d is None:
        return None
    w = d.get('workload', {})
    same = (w.get('N') == N and w.get('E') == E and not args.uniform
            and w.get('index_dtype') == args.index_dtype)
    return d.get('traffic_bytes_per_launch', {}).get(kernel) if same else None


# bench symbol -> substring of the device symbol rocprofv3 prints
_PMC_SYMBOLS = (('sage_fused_split_kernel<long,64>', 'sage_fused_split_kernel<long, 64'),
                ('sage_fused_split_kernel<long,32>', 'sage_fused_split_kernel<long, 32'),
                ('sage_fused_fwd_kernel<long,64>', 'sage_fused_fwd_kernel<long, 64'),
                ('sage_fused_fwd_kernel<long,32>', 'sage_fused_fwd_kernel<long, 32'),
                ('spmm_sum_rows<long,F=48>', 'spmm_sum_rows<long, 4, 16, 1, 0'),
                ('spmm_sum_rows_sparse<long,F=48>', 'spmm_sum_rows_sparse<long, 4, 16, 1'))
_live_traffic = None   # {bench symbol: HBM bytes per launch} from THIS run's PMC passes, or None


def live_pmc_traffic(args):
    """HBM bytes per launch of the aggregation kernels from two rocprofv3 PMC passes run NOW, on
    this box, over this very command (2 timed steps): `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    in separate passes with `--kernel-trace` only, as /opt/skills/guides/MI355X_MICROARCH.md
    prescribes; FETCH_SIZE doubled (gfx950 reports 64 B per 128-byte request), both in KiB.
    Returns {symbol: bytes} or None (no rocprofv3, already under a profiler, a pass failed or
    timed out: the committed profile then serves, and the line says which it was)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('PYGAMD_BENCH_CHILD') or shutil.which('rocprofv3') is None:
        return None
    if any('rocprof' in os.environ.get(k, '') for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES',
                                                        'HSA_TOOLS_LIB')):
        return None   # this process is being profiled itself
    sums = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='pygamd_pmc_', dir='/tmp')
        cmd = ['rocprofv3', '--pmc', ctr, '--kernel-trace', '-d', out, '-o', 'pmc',
               '--output-format', 'csv', '--', sys.executable, os.path.abspath(__file__),
               '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-side-figures',
               '--no-live-pmc', '--scale', str(args.scale), '--index-dtype', args.index_dtype]
        if args.uniform:
            cmd.append('--uniform')
        if args.arith is not None:
            cmd += ['--arith', args.arith]
        env = dict(os.environ, TMPDIR='/tmp', PYGAMD_BENCH_CHILD='1')
        try:
            res = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, timeout=300)
            if res.returncode != 0:
                return None
            vals = {}
            for f in glob.glob(os.path.join(out, '**', '*counter_collection*.csv'),
                               recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get('Counter_Name') != ctr:
                            continue
                        for sym, sub in _PMC_SYMBOLS:
                            if sub in r['Kernel_Name']:
                                vals.setdefault(sym, []).append(float(r['Counter_Value']))
            if not vals:
                return None
            sums[ctr] = {k: sum(v) / len(v) for k, v in vals.items()}
        except (OSError, ValueError, KeyError, subprocess.SubprocessError):
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return {k: (2.0 * sums['FETCH_SIZE'][k] + sums['WRITE_SIZE'][k]) * 1024.0
            for k in sums['FETCH_SIZE'] if k in sums['WRITE_SIZE']}


def measured_copy_bandwidth(dev, n_bytes: int = 1 << 30, reps: int = 5) -> dict:
    """GB/s (read + written bytes) of device-to-device copies of `n_bytes` on this box, timed with
    HIP events: SURVEY.md 8(d)'s secondary denominator for the HBM-bound kernels (the guide's
    figure for this chip: 6.29 TB/s for a float4 copy against the 8 TB/s spec).  Two copies:
    torch's `copy_` and the best schedule of the library's own streaming probe
    (pygamd_lab_copy: 4 / 8 loads in flight, plain / non-temporal, 4 / 8 / 16 workgroups per CU).
    Measurement probes, not part of the timed step."""
    from pytorch_geometric_amd import _lib
    src = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def rate(fn):
        fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return 2.0 * n_bytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    def probe(variant, per_cu):
        rc = _lib.load_lab().pygamd_lab_copy(src.data_ptr(), dst.data_ptr(), n_bytes, variant, per_cu,
                                        stream)
        if rc:
            raise RuntimeError(f'pygamd_lab_copy: {rc}')

    out = {'torch_copy': rate(lambda: dst.copy_(src))}
    best = None
    for variant in range(4):
        for per_cu in (4, 8, 16):
            r = rate(lambda: probe(variant, per_cu))
            if best is None or r > best[0]:
                best = (r, variant, per_cu)
    if not torch.equal(src, dst):
        raise RuntimeError('pygamd_lab_copy: copy differs from its source')
    read = None  # the read half alone (the dominant kernel reads 14 bytes for every byte it writes)
    for variant in range(4, 8):
        for per_cu in (4, 8, 16):
            r = rate(lambda: probe(variant, per_cu)) / 2.0
            if read is None or r > read[0]:
                read = (r, variant, per_cu)
    out['read'] = read[0]
    out['read_schedule'] = {'non_temporal': bool(read[1] & 1),
                            'loads_in_flight': 8 if read[1] & 2 else 4,
                            'workgroups_per_cu': read[2]}
    out['probe'] = best[0]
    out['probe_schedule'] = {'non_temporal': bool(best[1] & 1), 'loads_in_flight': 8 if best[1] & 2 else 4,
                             'workgroups_per_cu': best[2]}
    return out


def fused_algorithmic_bytes(info) -> float:
    """One-kernel SAGE layer (csrc/sage_fused.hip), forward or input gradient: the aggregation's
    SURVEY 8(d) bytes — E*(4F+b) + (N+1)*b, + N*4F only when the aggregated rows are stored
    (forward: the weight gradient reads them) — + the root rows N*4F + the output N*4Fo
    (+ N*Fo/8 of ReLU-mask bits read or written, + N*4Fo for the row-scaled second output)."""
    b, Fw, n = info['idx_bytes'], info['F'], info['n_rows']
    fg = info['fused_gemm']
    total = info['nnz'] * (4 * Fw + b) + (n + 1) * b + n * 4 * Fw + n * 4 * fg['Fo']
    if fg.get('save_agg', True):
        total += n * 4 * Fw
    total += n * 4 * ((fg['Fo'] + 31) // 32)
    if fg.get('scaled_copy'):
        total += n * 4 * fg['Fo']
    return float(total)


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
    reference travelled with the snapshot.

    Returns ``(json_dict, sample)``; ``sample`` = the inputs, the initial weights and the
    reference's loss / output / parameter gradients of one step, for `parity_at_cpu_scale`."""
    from pytorch_geometric_amd.datasets import products_like
    x, y, ei, c = products_like(seed=1, scale=scale)
    E = ei.size(1)
    g = torch.Generator().manual_seed(7)
    train_idx = torch.randperm(x.size(0), generator=g)[:max(int(0.0803 * x.size(0)), 1)]
    threads = torch.get_num_threads()
    base = {'unit': 'edges/s', 'cores': threads, 'host_cpus': os.cpu_count()}
    shape = f'{scale:g} x ogbn-products shape: N={x.size(0)}, E={E}'
    sample = {'x': x, 'y': y, 'ei': ei, 'train_idx': train_idx, 'classes': c}
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
        sample['state'] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        params = [(cv.lin_l.weight, cv.lin_l.bias, cv.lin_r.weight) for cv in model.convs]
        last = {}

        def step():
            t0 = time.perf_counter()
            for p in model.parameters():
                p.grad = None
            out = O.graphsage(x, ei, params)
            loss = F.cross_entropy(out[train_idx], y[train_idx])
            t1 = time.perf_counter()
            loss.backward()
            last['out'], last['loss'] = out.detach(), loss.detach()
            return t1 - t0, time.perf_counter() - t1

        t, tf, tb = _time_steps(step, steps)
        sample.update(out=last['out'], loss=last['loss'], kind='port',
                      grads={k: p.grad.clone() for k, p in model.named_parameters()})
        m64 = GraphSAGE(100, 256, num_layers=3, out_channels=c).double()
        m64.load_state_dict({k: v.double() for k, v in sample['state'].items()})
        p64 = [(cv.lin_l.weight, cv.lin_l.bias, cv.lin_r.weight) for cv in m64.convs]
        F.cross_entropy(O.graphsage(x.double(), ei, p64)[train_idx], y[train_idx]).backward()
        sample['grads64'] = {k: p.grad.clone() for k, p in m64.named_parameters()}
        return dict(base, value=3 * E / t, kind='port',
                    sample=(f'oracle/pyg_oracle.py graphsage fwd+bwd (index_select + scatter_add_ '
                            f'mean), {shape}, median of {steps} steps after 1 warm-up, '
                            f'{t * 1e3:.0f} ms/step')), sample
    torch.manual_seed(0)
    model = RefSAGE(100, 256, num_layers=3, out_channels=c)
    sample['state'] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    last = {}

    def make_step(graph):
        def step():
            t0 = time.perf_counter()
            model.zero_grad(set_to_none=True)
            out = model(x, graph)
            loss = F.cross_entropy(out[train_idx], y[train_idx])
            t1 = time.perf_counter()
            loss.backward()
            last['out'], last['loss'] = out.detach(), loss.detach()
            return t1 - t0, time.perf_counter() - t1
        return step

    t, tf, tb = _time_steps(make_step(ei), steps)
    # (the weights never change: any step's results are THE reference results of this sample)
    sample.update(out=last['out'], loss=last['loss'], kind='reference',
                  grads={k: p.grad.clone() for k, p in model.named_parameters()})
    # the same step in fp64 (one pass): the yardstick for the parameter gradients, whose fp32
    # values — reductions over N rows — differ between two correct fp32 implementations by more than
    # 1e-5 of their magnitude (the reference's own fp32 result sits 1e-4 from fp64 at this size)
    m64 = RefSAGE(100, 256, num_layers=3, out_channels=c).double()
    m64.load_state_dict({k: v.double() for k, v in sample['state'].items()})
    F.cross_entropy(m64(x.double(), ei)[train_idx], y[train_idx]).backward()
    sample['grads64'] = {k: p.grad.clone() for k, p in m64.named_parameters()}
    del m64
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
                                 f'torch.sparse.mm: fwd {tf2:.2f} s + bwd {tb2:.2f} s'}), sample


def parity_at_cpu_scale(sample, dev):
    """The CPU baseline's sample through the GPU stack (same initial weights): loss and output
    against the reference's at 1e-5 (north_star's contract; each error relative to the largest
    magnitude of the reference tensor), and EVERY parameter gradient against the fp64 result of the
    same step — a weight gradient is a sum over all N rows, and two correct fp32 implementations
    differ there by their summation orders: the reference's own fp32 gradients sit ~1e-4 from fp64
    at this size.  A gradient passes when the GPU's distance to fp64 is at most twice the
    reference's own (floor 2e-5); both distances and the GPU-vs-reference distance are reported."""
    from pytorch_geometric_amd.nn import GraphSAGE
    model = GraphSAGE(100, 256, num_layers=3, out_channels=sample['classes'])
    model.load_state_dict(sample['state'])
    model = model.to(dev)
    x, ei = sample['x'].to(dev), sample['ei'].to(dev)
    ti = sample['train_idx'].to(dev)
    out = model(x, ei)
    from pytorch_geometric_amd.nn.functional import cross_entropy as rows_cross_entropy
    loss = rows_cross_entropy(out, sample['y'].to(dev), ti)   # (the headline step's loss)
    loss.backward()
    torch.cuda.synchronize(dev)

    def rel(got, ref, scale=None):
        got, ref = got.detach().cpu().double(), ref.double()
        if not bool(torch.isfinite(got).all()):
            return float('inf')
        scale = ref if scale is None else scale
        return float((got - ref).abs().max() / max(float(scale.abs().max()), 1e-30))

    def rel2(got, ref):  # relative L2 distance: a single flipped ReLU unit barely moves it
        got, ref = got.detach().cpu().double(), ref.double()
        return float((got - ref).norm() / max(float(ref.norm()), 1e-30))

    errs = {'loss': rel(loss, sample['loss']), 'out': rel(out, sample['out'])}
    g64 = sample['grads64']
    gpu64 = {k: rel(p.grad, g64[k]) for k, p in model.named_parameters()}
    ref64 = {k: rel(sample['grads'][k], g64[k]) for k in gpu64}
    gpu64_l2 = {k: rel2(p.grad, g64[k]) for k, p in model.named_parameters()}
    ref64_l2 = {k: rel2(sample['grads'][k], g64[k]) for k in gpu64}
    gpuref = {k: rel(p.grad, sample['grads'][k], g64[k]) for k, p in model.named_parameters()}
    worst = max(gpu64, key=lambda k: gpu64[k] / max(2 * ref64[k], 2e-5))
    errs['param_grad_gpu_vs_fp64'] = max(gpu64.values())
    errs['param_grad_ref_vs_fp64'] = max(ref64.values())
    errs['param_grad_gpu_vs_ref'] = max(gpuref.values())
    errs['worst_param'] = worst
    errs['n_param_tensors'] = len(gpu64)
    errs['per_tensor_gpu_vs_fp64__ref_vs_fp64'] = {
        k: [float(f'{gpu64[k]:.2e}'), float(f'{ref64[k]:.2e}')] for k in gpu64}
    errs['per_tensor_relL2_gpu_vs_fp64__ref_vs_fp64'] = {
        k: [float(f'{gpu64_l2[k]:.2e}'), float(f'{ref64_l2[k]:.2e}')] for k in gpu64}
    errs['tol'] = {'loss': 1e-5, 'out': 1e-5,
                   'param_grad': 'NOT north_star\'s 1e-5 (sums over N rows behind two ReLUs: a '
                                 'hidden unit within fp32 rounding of 0 flips differently in any '
                                 'two fp32 evaluations, the reference\'s own included): per '
                                 'tensor, max-norm distance to fp64 <= max(2 x the reference\'s '
                                 'own, 2e-5), or — the max-norm is moved by single flips — relative '
                                 'L2 distance to fp64 <= max(1.5 x the reference\'s own, 1e-5); '
                                 'all four distances are printed'}
    errs['against'] = sample['kind']
    errs['ok'] = bool(errs['loss'] <= 1e-5 and errs['out'] <= 1e-5
                      and all(gpu64[k] <= max(2 * ref64[k], 2e-5)
                              or gpu64_l2[k] <= max(1.5 * ref64_l2[k], 1e-5) for k in gpu64))
    return {k: (float(f'{v:.3e}') if isinstance(v, float) else v) for k, v in errs.items()}


def parity_cached(dev):
    """`parity_at_cpu_scale` without a live CPU run: the GPU stack on the committed sample
    (tests/golden/golden_bench_sample_v1.pt, made by tests/golden/make_golden_bench_sample.py from
    the real reference at 1/128 of the products shape; inputs are regenerated from their seeds and
    checked against the stored checksums).  What rank 0 reports when world > 1."""
    from pytorch_geometric_amd.datasets import products_like
    from pytorch_geometric_amd.nn import GraphSAGE
    path = os.path.join(ROOT, 'tests', 'golden', 'golden_bench_sample_v1.pt')
    if not os.path.exists(path):
        return {'ok': None, 'against': 'no cached sample in this snapshot'}
    blob = torch.load(path, map_location='cpu', weights_only=False)
    x, y, ei, c = products_like(seed=1, scale=blob['scale'])
    g = torch.Generator().manual_seed(7)
    train_idx = torch.randperm(x.size(0), generator=g)[:max(int(0.0803 * x.size(0)), 1)]
    torch.manual_seed(0)
    model = GraphSAGE(100, 256, num_layers=3, out_channels=c)
    cs = {'x': float(x.double().abs().sum()), 'ei': float(ei.double().sum()),
          'state': sum(float(v.double().abs().sum()) for v in model.state_dict().values())}
    if any(abs(cs[k] - blob['checksums'][k]) > 1e-9 * abs(blob['checksums'][k]) for k in cs):
        return {'ok': None, 'against': 'cached sample: regenerated inputs differ (checksums)'}
    model = model.to(dev)
    ti = train_idx.to(dev)
    out = model(x.to(dev), ei.to(dev))
    from pytorch_geometric_amd.nn.functional import cross_entropy as rows_cross_entropy
    loss = rows_cross_entropy(out, y.to(dev), ti)   # (the headline step's loss)
    loss.backward()
    torch.cuda.synchronize(dev)

    def rel(got, ref, scale):
        got = got.detach().cpu().double()
        if not bool(torch.isfinite(got).all()):
            return float('inf')
        return float((got - ref.double()).abs().max() / max(scale, 1e-30))

    errs = {'loss': rel(loss, blob['loss'], float(blob['loss'].abs())),
            'out': rel(out[blob['rows'].to(dev)], blob['out_rows'], blob['out_absmax'])}
    g64 = blob['grads64_as_f32']
    gpu64 = {k: rel(p.grad, g64[k], float(g64[k].abs().max()))
             for k, p in model.named_parameters()}
    gpu64_l2 = {k: float((p.grad.detach().cpu().double() - g64[k].double()).norm()
                         / g64[k].double().norm()) for k, p in model.named_parameters()}
    ref64, ref64_l2 = blob['ref_vs_fp64'], blob['ref_vs_fp64_l2']
    errs['param_grad_gpu_vs_fp64'] = max(gpu64.values())
    errs['param_grad_ref_vs_fp64'] = max(ref64.values())
    errs['worst_param'] = max(gpu64, key=lambda k: gpu64[k] / max(2 * ref64[k], 2e-5))
    errs['per_tensor_gpu_vs_fp64__ref_vs_fp64'] = {
        k: [float(f'{gpu64[k]:.2e}'), float(f'{ref64[k]:.2e}')] for k in gpu64}
    errs['per_tensor_relL2_gpu_vs_fp64__ref_vs_fp64'] = {
        k: [float(f'{gpu64_l2[k]:.2e}'), float(f'{ref64_l2[k]:.2e}')] for k in gpu64}
    errs['tol'] = {'loss': 1e-5, 'out': 1e-5,
                   'param_grad': 'NOT north_star\'s 1e-5 (see parity_at_cpu_scale): per tensor, '
                                 'max-norm distance to fp64 <= max(2 x the reference\'s own, '
                                 '2e-5), or relative L2 distance <= max(1.5 x the reference\'s '
                                 'own, 1e-5)'}
    errs['against'] = (f'cached reference sample, {blob["scale"]:g} x products shape '
                       f'(N={blob["n"]}, E={blob["e"]}; 1024 output rows, every parameter '
                       f'gradient): tests/golden/golden_bench_sample_v1.pt')
    errs['ok'] = bool(errs['loss'] <= 1e-5 and errs['out'] <= 1e-5
                      and all(gpu64[k] <= max(2 * ref64[k], 2e-5)
                              or gpu64_l2[k] <= max(1.5 * ref64_l2[k], 1e-5) for k in gpu64))
    return {k: (float(f'{v:.3e}') if isinstance(v, float) else v) for k, v in errs.items()}


def _print_line(result: dict) -> None:
    print(json.dumps(result), flush=True)


# where run_minibatch() hands its result line (bench_configs.config4 collects it instead)
EMIT = _print_line


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
    if args.capture:
        return run_minibatch_captured(args, rank, world, dev, model, bucket, loader, fan, scale,
                                      N, E, t_gen)
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
        if dist.is_initialized():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean(force=True)
            e1.record()
            ar_events.append((e0, e1))
        opt.step()
        ne = b.num_sampled_edges  # layer l aggregates the edges of hops 0 .. L-1-l
        edges += sum(sum(ne[:len(ne) - l]) for l in range(len(ne)))
        return loss

    def fence():
        if dist.is_initialized():
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
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist.is_initialized():
        every = [torch.zeros_like(t[:1]) for _ in range(world)]
        dist.all_gather(every, t[:1].clone())
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        t[0] = tm[0]
    elapsed, total_edges = float(t[0]), float(t[1])
    if rank == 0:
        EMIT({
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
                           / max(len(ar_events), 1), 4) if ar_events else 0.0,
                       'per_rank_ms_per_step': {'min': round(min(per_rank_ms), 3),
                                                'max': round(max(per_rank_ms), 3),
                                                'ranks': [round(v, 3) for v in per_rank_ms]},
                       'graph_build_s': round(t_gen, 1),
                       'hbm_gb_allocated': round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                       'gemm': gemm_desc(False)}})


def run_minibatch_captured(args, rank, world, dev, model, bucket, loader, fan, scale, N, E, t_gen):
    """BASELINE config 4 with the WHOLE batch step — device-side neighbour sampling at static
    capacities, feature gather, the padded hop-aware GraphSAGE stack forward + backward, Adam — as
    ONE captured hipGraph (the eager path is bound by ~145 launches per batch).
    Only the seed ids change between replays (one device-to-device copy) plus a device-side RNG
    word.  With a process group over RCCL the gradient all-reduce is captured too (see below)."""
    import torch.distributed as dist

    from pytorch_geometric_amd.slots import run_slot_stack
    use_dist = dist.is_initialized()
    try:  # one multi-tensor kernel per step instead of ~18 foreach launches
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True)
    except (RuntimeError, TypeError, ValueError):
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, foreach=True)
    B = loader.batch_size
    seeds_buf = torch.zeros(B, dtype=torch.int64, device=dev)
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)   # batch counter ON the device: stamps
    loss_buf = torch.zeros((), device=dev)                   # the node claims and salts the draws

    def fwd_bwd():
        epoch.add_(1)
        b = loader.collate_slots(seeds_buf, epoch)
        bucket.zero_()
        out = run_slot_stack(model, b)
        loss = F.cross_entropy(out, b.y)
        loss.backward()
        loss_buf.copy_(loss.detach())

    def whole_step():
        fwd_bwd()
        opt.step()

    def plan():
        while True:
            for seeds, _ in loader._plan():
                if seeds.numel() == B:
                    yield seeds

    # With a process group over RCCL the gradient all-reduce and the optimizer step are recorded
    # INTO the graph as well (RCCL launches are stream-ordered kernels:
    # tests/test_gpu_nccl.py::test_rccl_all_reduce_and_adam_capture_into_one_hipgraph): an 8-rank
    # run replays ONE graph per batch, exactly like the one-GPU run.  gloo (the shared-GPU test
    # mode) cannot be captured: there the two stay behind the graph.
    in_graph = (use_dist and dist.get_backend() == 'nccl'
                and os.environ.get('PYGAMD_CAPTURE_COLLECTIVE', '1') != '0')

    def whole_step_dist():
        fwd_bwd()
        bucket.all_reduce_mean(force=True)
        opt.step()

    outside = use_dist and not in_graph   # all-reduce + Adam issued eagerly behind the graph
    body = whole_step_dist if in_graph else (fwd_bwd if use_dist else whole_step)
    it = plan()
    eager = os.environ.get('PYGAMD_CAPTURE', '1') == '0'  # the same static-shape step, eagerly
    # Round 6: the step is slots.SlotTrainer's — parameters in one flat buffer in the layer
    # kernels' layout, loss + its gradient and Adam as ONE launch each, no per-parameter
    # concatenation / transpose / accumulation launch (PYGAMD_SLOT_TRAINER=0: the round-5 step
    # through autograd + F.cross_entropy + torch's fused Adam, for an A/B on the same box).
    use_trainer = os.environ.get('PYGAMD_SLOT_TRAINER', '1') != '0'
    piped = False
    if use_trainer:
        from pytorch_geometric_amd.slots import SlotTrainer
        # (PYGAMD_SLOT_PIPELINE=0: the next batch is drawn in front of its own training instead of
        # beside the previous one's)
        piped = os.environ.get('PYGAMD_SLOT_PIPELINE', '1') != '0'
        trainer = SlotTrainer(model, loader, lr=1e-3, capture=not eager, pipeline=piped,
                              collective_in_graph=in_graph if use_dist else None)
        seeds_buf, epoch, loss_buf = trainer.seeds, trainer.epoch, trainer.loss
        seeds_buf.copy_(next(it))
        captured = trainer.step          # (records itself on the first call)
        captured()
        outside = False                  # (a gloo group: the trainer reduces behind its graph)
    else:
        seeds_buf.copy_(next(it))
        from pytorch_geometric_amd.hipgraph import CapturedStep
        if eager:
            captured = body                                # (per-kernel times for a profile)
            for _ in range(3):
                captured()
        else:
            captured = CapturedStep(body, warmup=3)
    ar_events = []

    def step():
        seeds_buf.copy_(next(it))
        captured()
        if outside:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean(force=True)
            e1.record()
            ar_events.append((e0, e1))
            opt.step()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    del ar_events[:]
    timed_seeds = []
    first_epoch = int(epoch.item()) + 1
    _next = it.__next__

    def next_logged():
        sd = _next()
        timed_seeds.append(sd)
        return sd

    trace = [] if os.environ.get('PYGAMD_STEP_TRACE') == '1' else None  # (diagnosis only)

    def step():  # (the timed twin of the warm-up step: it also remembers its seeds)
        ts = [time.perf_counter()] if trace is not None else None

        def mark():
            if ts is not None:
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter())

        seeds_buf.copy_(next_logged())
        mark()
        captured()
        mark()
        if outside:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean(force=True)
            e1.record()
            ar_events.append((e0, e1))
            mark()
            opt.step()
            mark()
        if ts is not None:
            trace.append([round((b - a) * 1e3, 3) for a, b in zip(ts, ts[1:])])

    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if trace is not None and rank == 0:
        print('[step trace, ms: seeds copy | step | all-reduce | optimizer]', trace[:8],
              file=sys.stderr, flush=True)
    assert torch.isfinite(loss_buf).item()
    # what the timed batches contained, recounted OUTSIDE the timed region (the captured step keeps
    # no statistics): the sampler is a pure function of (seeds, epoch) — the draws depend on them
    # only — so a SECOND sampler with a fresh claim map, fed the same epochs in the same order,
    # reproduces every batch (the first one's map already holds later epochs)
    from pytorch_geometric_amd.slots import SlotSampler
    plan = loader._slots.plan
    again = SlotSampler(loader._slots.colptr, loader._slots.row, loader.num_nodes, plan,
                        seed=loader._slots.seed)
    hop_edges = torch.zeros(len(fan), dtype=torch.int64, device=dev)
    hop_nodes = torch.zeros(len(fan), dtype=torch.int64, device=dev)
    ep = torch.zeros(1, dtype=torch.int64, device=dev)
    for i, sd in enumerate(timed_seeds):
        ep.fill_(first_epoch + i)
        sb = again.sample(sd, ep)
        for h in range(len(fan)):
            hop_edges[h] += (sb.src_g[plan.ebase(h):plan.ebase(h + 1)] >= 0).sum()
            hop_nodes[h] += (sb.node_g[plan.bases[h + 1]:plan.bases[h + 2]] >= 0).sum()
    ne = [int(v) for v in hop_edges.tolist()]
    nn_ = [int(v) for v in hop_nodes.tolist()]
    Lh = len(ne)
    edges = sum(sum(ne[:Lh - l]) for l in range(Lh))   # layer l aggregates hops 0 .. L-1-l
    # algorithmic HBM bytes of the aggregation + feature gather: per aggregated
    # edge one source row (4 F) + one index, per gathered node one 512-byte feature row
    widths = [128, 256, 256]
    agg_bytes = sum(sum(ne[:Lh - l]) * (4 * widths[l] + 8) for l in range(Lh))
    gather_bytes = (B * args.steps + sum(nn_)) * 128 * 4
    t = torch.tensor([elapsed, float(edges)], dtype=torch.float64, device=dev)
    per_rank_ms = [elapsed / args.steps * 1e3]
    if use_dist:
        every = [torch.zeros_like(t[:1]) for _ in range(world)]
        dist.all_gather(every, t[:1].clone())
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        t[0] = tm[0]
    elapsed, total_edges = float(t[0]), float(t[1])
    if rank == 0:
        caps = [B]
        for k in fan:
            caps.append(caps[-1] * k)
        EMIT({
            'metric': 'edges/sec (fwd+bwd) 3-layer SAGE + NeighborLoader [15,10,5], '
                      'papers100M shape (BASELINE config 4, informational)',
            'value': total_edges / elapsed, 'unit': 'edges/s',
            'n_gpus': dist.get_world_size() if use_dist else 1,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'GraphSAGE(128->256->256->172) mini-batch training, batch '
                                   f'{B} seeds/rank, fan-out {fan}, hop-aware (trim_to_layer) '
                                   f'stack on STATIC-shape SLOT batches (csrc/minibatch.hip; block '
                                   f'capacities {caps}; one fused-layer launch per layer forward, '
                                   f'dgrad GEMMs + transposed SpMM backward'
                                   + ('; slots.SlotTrainer: flat parameters, one-launch loss and '
                                      'Adam' + ('; the NEXT batch is drawn (sampling + gather) as '
                                                'a parallel branch of the graph that trains on '
                                                'the current one' if piped else '')
                                      if use_trainer else '') + '), '
                                   f'synthetic papers100M shape x {scale:g} (N={N}, E={E}) '
                                   f'replicated per GPU',
                       'captured': (('forward + loss + backward' if piped else
                                     'sampling + gather + forward + backward') +
                                    (' + RCCL all-reduce + Adam' if in_graph else
                                     '' if use_dist else ' + Adam') + ' = one hipGraph per batch' +
                                    ('; sampling + gather of the NEXT batch = a second recording '
                                     'replayed beside it on a second stream' if piped else '')),
                       'launches_per_batch': (38 if use_trainer else 60),
                       'parallelism': f'dp{world} (seed sharding, one flat-bucket '
                                      f'all-reduce/step)',
                       'allreduce_ms_per_step': round(
                           sum(a.elapsed_time(b) for a, b in ar_events)
                           / max(len(ar_events), 1), 4) if ar_events else 0.0,
                       'per_rank_ms_per_step': {'min': round(min(per_rank_ms), 3),
                                                'max': round(max(per_rank_ms), 3),
                                                'ranks': [round(v, 3) for v in per_rank_ms]},
                       'real_edges_per_batch_per_hop': [round(v / args.steps, 1) for v in ne],
                       'new_nodes_per_batch_per_hop': [round(v / args.steps, 1) for v in nn_],
                       'graph_build_s': round(t_gen, 1),
                       'hbm_gb_allocated': round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                       'gemm': gemm_desc(False)},
            'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': 8000.0,
                         'achieved': round((agg_bytes + gather_bytes) / elapsed / 1e9, 1),
                         'frac': round((agg_bytes + gather_bytes) / elapsed / 8e12, 4),
                         'traffic': None,
                         'kernel': 'whole captured batch step (aggregation + feature gather '
                                   'bytes over the step time: the step is latency-bound, not '
                                   'bandwidth-bound)',
                         'algorithmic_bytes_per_batch': round(
                             (agg_bytes + gather_bytes) / args.steps)}})


def gemm_desc(tuned: bool) -> str:
    from pytorch_geometric_amd.nn.models import _fused_sage
    if _fused_sage.GEMM_BACKEND == 'own':
        from pytorch_geometric_amd import get_gemm_mode
        if get_gemm_mode() == 'split':
            return ('pytorch_geometric_amd/csrc/gemm.hip + sage_fused.hip, arithmetic "split" '
                    '(the default since round 4): every fp32 operand as the exact sum of 3 bf16 '
                    'terms, the 6 leading cross products on v_mfma_f32_32x32x16_bf16, fp32 '
                    'accumulation — fp32 inputs and outputs, error vs fp64 <= the fp32 '
                    'instruction\'s on the same inputs (tests/test_gpu_split_accept.py, '
                    'tests/test_gpu_gemm.py); stand-alone forward / dgrad / wgrad kernels AND the '
                    'transform phase of the one-kernel layers')
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
    if world > 1 or getattr(args, 'init_dist', False):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        # PYGAMD_BENCH_SHARE_GPU=1 (tests only, never a measurement): every rank on GPU 0, the
        # collectives over gloo on device tensors — RCCL refuses two ranks on one device, and a
        # one-GPU box is all the builder has: this is how the N > 1 compute path (per-rank graphs,
        # parameter broadcast, bucket all-reduce, max-over-ranks timing, the parity leg) runs on
        # hardware before the driver's 8-GPU node does
        share = has_gpu and os.environ.get('PYGAMD_BENCH_SHARE_GPU') == '1'
        if share:
            local_rank = 0
        if has_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl' if has_gpu and not share else 'gloo', rank=rank,
                                world_size=world)
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
    ap.add_argument('--arith', choices=['split', 'fp32'], default=None,
                    help="arithmetic of the dense transforms (default: the library's, 'split' "
                         "unless PYGAMD_GEMM_MODE says otherwise): 'split' = 3 x bf16 terms per "
                         "fp32 operand, 6 bf16 matrix products, fp32 accumulation; 'fp32' = the "
                         "exact fp32 matrix instruction")
    ap.add_argument('--torch-loss', action='store_true',
                    help='the loss of the training rows as F.cross_entropy(out[train_idx], ...) '
                         '(seven ATen launches) instead of the one-pass '
                         'pytorch_geometric_amd.nn.functional.cross_entropy')
    ap.add_argument('--no-side-figures', action='store_true',
                    help='full-batch mode: skip the short re-timings printed beside the headline '
                         '(exact fp32 instruction, dense loss)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the other_configs block (BASELINE configs 1/3/4/5 and the '
                         'reference-class headline, bench_configs.py; after the timed region)')
    ap.add_argument('--no-live-pmc', action='store_true',
                    help='full-batch mode: do not run the two rocprofv3 --pmc passes that give '
                         'roofline.traffic for THIS run (the committed profile serves instead)')
    ap.add_argument('--no-tuned-gemm', action='store_true',
                    help='use the default rocBLAS/hipBLASLt heuristics instead of the shipped '
                         'TunableOp table')
    ap.add_argument('--capture', action='store_true',
                    help='minibatch mode: sampling + gather + forward + backward + Adam of a batch '
                         'as ONE hipGraph on static-shape (padded) batches')
    ap.add_argument('--init-dist', action='store_true',
                    help='initialise torch.distributed (nccl = RCCL) also for ONE rank and run the '
                         'broadcast / all-reduce / barrier collectives on it (exercises the '
                         'multi-GPU code path on a single-GPU box)')
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
    if args.arith is not None:
        pga.set_gemm_mode(args.arith)
    tuned = False
    # (TunableOp serves the library GEMMs of the sampled-batch stack's small blocks only: the
    # full-batch step launches no library GEMM)
    if not args.no_tuned_gemm and args.mode == 'minibatch':
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
    try:  # one multi-tensor launch per step instead of ~10 foreach launches (same update)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    except (RuntimeError, TypeError, ValueError):
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    # like every full-batch PyG example, the loss is taken on the training split only
    # (ogbn-products: 196,615 of 2,449,029 nodes = 8 %)
    g = torch.Generator().manual_seed(7 + rank)
    train_idx = torch.randperm(N, generator=g)[:max(int(0.0803 * N), 1)].to(dev)
    y_train = y[train_idx]
    # stored entries that point at a training node: the rows of the loss gradient the transposed
    # aggregation of the output layer has to read (the others are zero and skipped, src_bits)
    live_nnz = int(torch.bincount(ei[1].long(), minlength=N)[train_idx].sum().item())

    ar_events = []  # (start, end) HIP events around the gradient all-reduce, timed steps only

    # The loss of the training rows: pytorch_geometric_amd.nn.functional.cross_entropy reads the
    # selected rows of `out` in ONE pass (loss + its gradient; round 6) where
    # F.cross_entropy(out[train_idx], y_train) is seven ATen launches (0.75 ms here: its nll_loss
    # reductions run on one workgroup).  --torch-loss keeps the ATen form; the default line prints
    # that step beside the headline (ms_per_step_torch_cross_entropy).
    from pytorch_geometric_amd.nn.functional import cross_entropy as rows_cross_entropy
    fused_loss = [not args.torch_loss]

    def step():
        bucket.zero_()
        out = model(x, ei)
        if fused_loss[0]:
            loss = rows_cross_entropy(out, y, train_idx)
        else:
            loss = F.cross_entropy(out[train_idx], y_train)
        loss.backward()
        if dist.is_initialized():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            # (the current stream waits for RCCL's stream before e1)
            bucket.all_reduce_mean(force=True)
            e1.record()
            ar_events.append((e0, e1))
        opt.step()
        return loss

    def fence():
        if dist.is_initialized():
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
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist.is_initialized():
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)  # each rank's own clock: a straggler shows up by name
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = 3.0 * E * world / (elapsed / args.steps)

    allreduce_ms = (sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1)
                    if ar_events else 0.0)
    n_pg = dist.get_world_size() if dist.is_initialized() else 1

    # ---- roofline: every launch of this repo's kernels in the timed region carries HIP events
    # (aggregation, one-kernel layers, GEMMs); the DOMINANT kernel is the device symbol with the
    # largest summed time ----
    def symbol(info):
        if info.get('kind') == 'gemm':
            # (csrc/gemm.hip: the weight gradient has its own kernel in the split arithmetic)
            # (and, from 32 k rows, a narrow g — N <= 96, here the classifier layer's 2 x 48 columns
            # against ONE x operand — its one-tile-per-workgroup form; the dispatch rule of
            # pygamd_linear_wgrad2 restated for the label only)
            tn = 'gemm_tn_split_kernel' if pga.get_gemm_mode() == 'split' else 'gemm_tn_kernel'
            if (tn == 'gemm_tn_split_kernel' and info['op'] == 'wgrad' and info.get('N', 1 << 30) <= 96
                    and info.get('N', 1) % 4 == 0 and info.get('K', 1) % 4 == 0
                    and info.get('M', 0) >= 32768 and not info.get('x2')):
                tn = 'gemm_tn_skinny_kernel'
            return {'wgrad': tn, 'dgrad': 'gemm_nt_kernel', 'forward': 'gemm_nt_kernel'}[info['op']]
        lpr = 4
        while lpr < 64 and lpr * 4 < info['F']:
            lpr <<= 1
        it = 'long' if info['idx_bytes'] == 8 else 'int'
        if info.get('fused_gemm'):
            # (csrc/sage_fused.hip: the production kernel of the current arithmetic; the lab
            # schedules of include/pyg_amd_lab.h when PYGAMD_FUSED_VARIANT asks for one)
            prod = ('sage_fused_split_kernel' if pga.get_gemm_mode() == 'split'
                    else 'sage_fused_fwd_kernel')
            name = {0: prod, 1: 'sage_fused_probe_kernel', 2: 'sage_fused_stream_kernel',
                    5: 'sage_fused_split_kernel', 6: 'sage_fused_fwd_kernel'}.get(
                        _native.SAGE_FUSED_VARIANT, 'sage_fused_spec_kernel')
            return f'{name}<{it},{lpr}>'
        if info.get('src_bits'):
            return f'spmm_sum_rows_sparse<{it},F={info["F"]}>'
        return f'spmm_sum_rows<{it},F={info["F"]}>'

    def alg_bytes(info):
        if info.get('src_bits'):
            info = dict(info, live_nnz=live_nnz)
        return (fused_algorithmic_bytes(info) if info.get('fused_gemm')
                else spmm_algorithmic_bytes(info))

    by_sym = {}
    for info, ev0, ev1 in sink:
        by_sym.setdefault(symbol(info), []).append((info, ev0.elapsed_time(ev1)))
    tot_ms = {k: sum(ms for _, ms in v) for k, v in by_sym.items()}
    agg_syms = [k for k in by_sym if not k.startswith('gemm')]
    dom = max(agg_syms, key=tot_ms.get) if agg_syms else None
    if tot_ms and max(tot_ms, key=tot_ms.get) != dom:
        dom_any = max(tot_ms, key=tot_ms.get)
    else:
        dom_any = dom

    def hbm_entry(sym):
        items = by_sym[sym]
        ms = sum(t for _, t in items) / len(items)
        byts = sum(alg_bytes(i) for i, _ in items) / len(items)
        strict = sum(spmm_algorithmic_bytes(dict(i, accumulate=False, relu_mask=False,
                                                 relu_bits=False)) for i, _ in items) / len(items)
        ach = byts / (ms * 1e-3) / 1e9
        e = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
             'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': pmc_traffic(args, N, E, sym),
             'kernel': sym, 'launches_per_step': round(len(items) / args.steps, 2),
             'avg_launch_ms': round(ms, 4), 'ms_per_step': round(tot_ms[sym] / args.steps, 3),
             'algorithmic_bytes_per_launch': byts,
             'frac_strict_8d': round(strict / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        fg = [i['fused_gemm'] for i, _ in items if i.get('fused_gemm')]
        if fg:
            flops = sum(2.0 * i['n_rows'] * i['fused_gemm']['K'] * i['fused_gemm']['Fo']
                        for i, _ in items) / len(items)
            e['mfma_frac_of_157.3'] = round(flops / (ms * 1e-3) / 1e12 / 157.3, 4)
        return e

    # HBM traffic of this run: two PMC passes over this command in child processes (rank 0 of a
    # one-GPU run, not for the short profiling invocations)
    global _live_traffic
    if (rank == 0 and world == 1 and not args.no_live_pmc and not args.no_side_figures
            and not args.no_cpu_baseline):
        _live_traffic = live_pmc_traffic(args)
    roofline = hbm_entry(dom) if dom else {}
    if dom:
        roofline['traffic_source'] = (
            'live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, '
            '--kernel-trace only) over this command with 2 steps, run by this process; '
            '(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch'
            if _live_traffic is not None else
            'committed profile (profiles/r06_pmc_bench.json, same command and workload)')
    cp = None
    if dom and rank == 0:  # the same launch against what plain device copies reach on THIS box
        try:
            cp = measured_copy_bandwidth(dev)
        except RuntimeError as exc:  # (a side figure: never at the price of the bench line)
            roofline['copy_bandwidth'] = {'error': str(exc)[:200]}
    if cp is not None:
        copy_gbs = max(cp['torch_copy'], cp['probe'])
        roofline['copy_bandwidth'] = {
            'measured': round(copy_gbs, 1), 'unit': 'GB/s',
            'what': 'fastest device-to-device copy of 1 GiB on this box (read + written bytes), '
                    'HIP events, this run',
            'torch_copy': round(cp['torch_copy'], 1), 'lab_copy': round(cp['probe'], 1),
            'lab_schedule': cp['probe_schedule'],
            'lab_read_only': round(cp['read'], 1), 'lab_read_schedule': cp['read_schedule'],
            'frac_of_read': round(roofline['achieved'] / cp['read'], 4),
            'frac_of_copy': round(roofline['achieved'] / copy_gbs, 4)}
    roofline['share_of_step'] = round(tot_ms.get(dom, 0.0) / (elapsed * 1e3), 4) if dom else None
    roofline['others'] = {k: round(tot_ms[k] / args.steps, 3) for k in sorted(tot_ms) if k != dom}
    if dom_any != dom:  # (a GEMM symbol leads: say so; the HBM entry above stays the aggregation)
        roofline['largest_symbol'] = dom_any
    second = sorted((k for k in agg_syms if k != dom), key=tot_ms.get, reverse=True)[:1]
    if second:
        e2 = hbm_entry(second[0])
        roofline['second'] = {k: e2[k] for k in ('kernel', 'frac', 'achieved', 'avg_launch_ms',
                                                 'ms_per_step', 'traffic')}
    # whole-step figures: every aggregation launch against HBM, every stand-alone GEMM against the
    # fp32 matrix peak (the flops inside the one-kernel layers are listed, not priced twice)
    agg_ms = sum(tot_ms[k] for k in agg_syms) / args.steps
    agg_bytes = sum(alg_bytes(i) for k in agg_syms for i, _ in by_sym[k]) / args.steps
    gemm_items = [(i, ms) for k in by_sym if k.startswith('gemm') for i, ms in by_sym[k]]
    gemm_ms = sum(ms for _, ms in gemm_items) / args.steps
    gemm_flop = sum(2.0 * i['M'] * i['N'] * i['K'] for i, _ in gemm_items) / args.steps
    fused_flop = sum(2.0 * i['n_rows'] * i['fused_gemm']['K'] * i['fused_gemm']['Fo']
                     for k in agg_syms for i, _ in by_sym[k] if i.get('fused_gemm')) / args.steps
    roofline['step'] = {
        'agg_ms': round(agg_ms, 3), 'agg_GB': round(agg_bytes / 1e9, 2),
        'agg_frac_of_hbm_peak': round(agg_bytes / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if agg_ms > 0 else None,
        'gemm_ms': round(gemm_ms, 3), 'gemm_GFLOP': round(gemm_flop / 1e9, 1),
        'gemm_frac_of_157.3TF': round(gemm_flop / (gemm_ms * 1e-3) / 1e12 / 157.3, 4)
        if gemm_ms > 0 else None,
        'GFLOP_inside_fused_layers': round(fused_flop / 1e9, 1),
        'other_ms': round(ms_per_step - agg_ms - gemm_ms, 3)}

    # ---- figures printed BESIDE the headline (never the headline): the same step with the exact
    # fp32 matrix instruction everywhere, and with the loss gradient treated as dense (no zero-row
    # skipping in the output layer's transposed aggregation).  Every rank runs them (the step
    # contains the all-reduce); short: 2 + 5 steps each.
    def quick_ms(n_warm=2, n=5):
        for _ in range(n_warm):
            step()
        fence()
        q0 = time.perf_counter()
        for _ in range(n):
            step()
        fence()
        return (time.perf_counter() - q0) / n * 1e3

    side = {}
    if not args.no_side_figures:
        from pytorch_geometric_amd.nn.models import _fused_sage
        mode = pga.get_gemm_mode()
        if mode != 'fp32':
            pga.set_gemm_mode('fp32')
            side['ms_per_step_exact_fp32_instruction'] = round(quick_ms(), 3)
            pga.set_gemm_mode(mode)
        if _fused_sage.SPARSE_GRAD:
            _fused_sage.SPARSE_GRAD = False
            side['dense_loss_ms_per_step'] = round(quick_ms(), 3)
            _fused_sage.SPARSE_GRAD = True
        if fused_loss[0]:
            fused_loss[0] = False
            side['ms_per_step_torch_cross_entropy'] = round(quick_ms(), 3)
            fused_loss[0] = True

    arithmetic = (
        'fp32 in / fp32 out; products as 3 x bf16 split (x = x1 + x2 + x3 exactly), the 6 leading '
        'cross terms on v_mfma_f32_32x32x16_bf16, fp32 accumulation; error vs fp64 <= the exact '
        'fp32 instruction\'s on the same inputs at the headline shapes '
        '(tests/test_gpu_split_accept.py); ms_per_step_exact_fp32_instruction = the same step on '
        'v_mfma_f32_32x32x2_f32' if pga.get_gemm_mode() == 'split'
        else 'v_mfma_f32_32x32x2_f32 (exact fp32 products and sums, bitwise an fmaf chain)')

    if rank == 0:
        result = {
            'metric': 'edges/sec (fwd+bwd) 3-layer SAGE, ogbn-products shape',
            'value': value, 'unit': 'edges/s', 'n_gpus': n_pg, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'arithmetic': arithmetic, **side,
            'config': {
                'workload': (f'GraphSAGE(100->256->256->{num_classes}, mean aggr) full-batch '
                             f'fwd+bwd(CE on an 8% train split' + (', taken by the one-pass pytorch_geometric_amd.nn.functional.cross_entropy' if fused_loss[0] else ', F.cross_entropy on the gathered rows') + ')+Adam on a synthetic ogbn-products-shaped graph per GPU '
                             f'(N={N}, E={E}, {"uniform" if args.uniform else "power-law"} '
                             f'degrees, {args.index_dtype} edge_index, fp32 features)'),
                'edges_per_step_per_gpu': 3 * E, 'scale': args.scale,
                'parallelism': f'dp{world} (graph replicas, one flat-bucket all-reduce/step)',
                'allreduce_ms_per_step': round(allreduce_ms, 4),
                'per_rank_ms_per_step': {'min': round(min(per_rank_ms), 3),
                                         'max': round(max(per_rank_ms), 3),
                                         'ranks': [round(v, 3) for v in per_rank_ms]},
                'process_group': dist.get_backend() if dist.is_initialized() else None,
                'graph_gen_s': round(t_gen, 1),
                # (short copies of the top-level `arithmetic` note and of the side figures: the
                # driver's record keeps scalars and the first 120 characters of strings in here)
                'arithmetic': ('3xbf16 split, 6 products, fp32 accumulate; err vs fp64 <= fp32 '
                               'instruction (tests/test_gpu_split_accept.py)'
                               if pga.get_gemm_mode() == 'split' else 'v_mfma_f32_32x32x2_f32'),
                **side,
                'gemm': gemm_desc(tuned),
                'schedule': 'layers 1-2 forward AND layer 2 input gradient as ONE kernel each '
                            '(gather -> LDS -> MFMA); 256->47 layer transforms first, '
                            'aggregates at width 48; ReLU backward / bias grads / 1/deg in '
                            'epilogues',
            },
            'roofline': roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            result['cpu_baseline'], sample = cpu_baseline(args.cpu_scale)
            result['parity_at_cpu_scale'] = parity_at_cpu_scale(sample, dev)
            # (the leg an N > 1 line carries instead, run here too so that it is exercised on
            # every round's single-GPU box)
            result['parity_cached_sample'] = parity_cached(dev)
            if not args.no_other_configs and not args.no_side_figures:
                # BASELINE configs 1 / 3 / 4 / 5 and this step written with the REFERENCE's
                # GraphSAGE + install(), after the timed region (bench_configs.py); scalars only
                import bench_configs
                result['other_configs'] = bench_configs.run(
                    dev, headline=(x, ei, train_idx, y_train, num_classes,
                                   side.get('ms_per_step_torch_cross_entropy', ms_per_step)))
        elif not args.no_cpu_baseline:
            # N > 1: no live CPU run of the reference (rank 0 only, the other ranks wait at the
            # barrier below); the GPU leg against the committed reference sample
            result['cpu_baseline'] = None
            result['parity_at_cpu_scale'] = parity_cached(dev)
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""The LITERAL north_star path at the products shape: gather rows of x on
edge_index[0] -> (identity / weighted message) -> scatter-{sum, mean, max} onto edge_index[1], and
its backward (gather of grad_out on edge_index[1] -> scatter-add onto edge_index[0]) — the route
every user-defined `message()` takes (nn/conv/message_passing.py:263-290, utils/_scatter.py:68-100),
with the [E, F] message tensor materialised in HBM.  Sorted index = the destination-sorted edge
list (scatter = one segment reduction per destination, no atomics); unsorted = the edge list as
generated (fp32 atomics / compare-and-swap loops).
Algorithmic bytes: gather E*(4F + b) read + E*4F written; scatter E*(4F + b) read + N*4F written.
Usage: python scripts/unfused_probe.py [--scale s] > profiles/r05_unfused_propagate.md"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_geometric_amd as pga  # noqa: E402
from pytorch_geometric_amd import _native  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--widths', default='100,256')
args = ap.parse_args()
dev = torch.device('cuda:0')
PEAK = 8000.0


def timeit(fn, reps=3):
    try:
        return _timeit(fn, reps)
    except Exception as exc:  # a combination the kernels do not take: say so in the table
        print(f'<!-- {type(exc).__name__}: {exc} -->')
        return float('nan')


def _timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


_, _, ei, _ = products_like(seed=1, scale=args.scale)
x0_rows = int(ei.max()) + 1
ei = ei.to(dev)
N, E = x0_rows, ei.size(1)
h = pga.EdgeIndex(ei, (N, N))
fwd = h.by_dst()
dst_sorted = _native.ptr2index(fwd.ptr, E)       # destination of every sorted slot
src_sorted = fwd.idx                             # its source
src_u, dst_u = ei[0].contiguous(), ei[1].contiguous()
b = 8
from pytorch_geometric_amd import _functions  # noqa: E402
from pytorch_geometric_amd.utils import scatter as u_scatter  # noqa: E402
_t0, _t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_t0.record()
_functions._sorted_scatter_plan(dst_u, N)
_t1.record()
torch.cuda.synchronize()
SORT_MS = _t0.elapsed_time(_t1)
print(f'# The unfused propagate path at the products shape (N = {N}, E = {E}, int64 indices, fp32), '
      f'one MI355X\n')
print('`python scripts/unfused_probe.py`: every launch timed alone (3 repetitions after one '
      'warm-up), algorithmic bytes as in the docstring, fraction of the 8 TB/s HBM peak.\n')
print(f'One-time cost of the cached plan `utils.scatter` builds for the unsorted index (stable radix '
      f'sort of E int64 keys + pointer + hub plan + one host read): {SORT_MS:.1f} ms.\n')
print('| F | step | index | kernel(s) | ms | GB | GB/s | frac of 8 TB/s |')
print('|---:|---|---|---|---:|---:|---:|---:|')
g = torch.Generator(device=dev).manual_seed(0)
for F in [int(v) for v in args.widths.split(',')]:
    x = torch.randn(N, F, device=dev, generator=g)
    go = torch.randn(N, F, device=dev, generator=g)
    w_edge = torch.rand(E, device=dev, generator=g)
    gb_gather = (E * (4 * F + b) + E * 4 * F) / 1e9
    gb_scatter = (E * (4 * F + b) + N * 4 * F) / 1e9

    def row(step, index, kern, ms, gb):
        print(f'| {F} | {step} | {index} | {kern} | {ms:.2f} | {gb:.1f} | {gb / ms * 1e3:.0f} | '
              f'{gb / ms * 1e3 / PEAK:.3f} |', flush=True)

    for name, src, dst in (('sorted by destination', src_sorted, dst_sorted),
                           ('unsorted', src_u, dst_u)):
        srt = name.startswith('sorted')
        msg = _native.gather_rows(x, src)
        row('gather x[edge_index[0]]', name, 'gather_rows_kernel',
            timeit(lambda: _native.gather_rows(x, src)), gb_gather)
        if srt:
            t = timeit(lambda: msg.mul_(w_edge.view(-1, 1)))
            row('message x_j * w_e (in place)', name, 'ATen mul (user message)', t,
                (2 * E * 4 * F + E * 4) / 1e9)
        for reduce in ('sum', 'mean', 'max'):
            if srt:
                fn = (lambda r=reduce: _native.spmm_csr(fwd.ptr, None, msg, r, n_rows=N))
                kern = 'spmm_sum_rows<IDENT>' if reduce != 'max' else 'spmm_minmax_rows_plain<IDENT>'
            else:
                fn = (lambda r=reduce: _native.scatter_rows(msg, dst, N, r))
                kern = f'scatter_rows_kernel<{reduce}> (atomics)'
            row(f'scatter-{reduce} onto edge_index[1]', name, kern, timeit(fn), gb_scatter)
            if not srt:
                # what `utils.scatter` does since round 4 for a large unsorted index: one cached
                # radix sort of the index, then a gather-SpMM over the sorted groups
                row(f'scatter-{reduce} via utils.scatter (cached sort + segment reduction)', name,
                    'spmm rows kernels, col = sort permutation',
                    timeit(lambda r=reduce: u_scatter(msg, dst, 0, N, r)), gb_scatter + E * 8 / 1e9)
        # backward of scatter-sum: grad_msg = grad_out[edge_index[1]]; of the gather: scatter-add
        # of grad_msg onto edge_index[0] (never sorted when the list is destination-sorted)
        gmsg = _native.gather_rows(go, dst)
        row('bwd: gather grad_out[edge_index[1]]', name, 'gather_rows_kernel',
            timeit(lambda: _native.gather_rows(go, dst)), gb_gather)
        row('bwd: scatter-add onto edge_index[0]', name + ' (sources are never sorted)',
            'scatter_rows_kernel<sum> (atomics)',
            timeit(lambda: _native.scatter_rows(gmsg, src, N, 'sum')), gb_scatter)
        del msg, gmsg
    # the fused route for comparison: one SpMM launch, no [E, F] tensor
    t = timeit(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, x, 'mean', n_rows=N, hub=fwd.hub))
    gb = (E * (4 * F + b) + (N + 1) * b + N * 4 * F) / 1e9
    row('FUSED gather + mean (message_and_aggregate)', 'sorted by destination', 'spmm_sum_rows', t,
        gb)
    del x, go, w_edge
    torch.cuda.empty_cache()

"""Own NT / TN kernels (csrc/gemm.hip: 64 x 64 tiles + split over the reduction, 256-row splits of
M) against the library (torch.mm = rocBLAS / hipBLASLt) at the row counts of the sampled-batch
blocks, the Cora shape and the FB15k-237 node count.
Usage: python scripts/small_gemm_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [(1024, 512, 172), (1024, 512, 256), (2708, 1433, 16), (2708, 16, 7), (14541, 100, 32),
          (14541, 256, 256), (15360, 256, 512), (16384, 512, 256), (16384, 256, 512)]


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


g = torch.Generator(device=dev).manual_seed(0)
print('shape (M, K, N): forward own fp32 / own split / library us | dgrad ... | wgrad ...')
for M, K, N in SHAPES:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g)
    go = torch.randn(M, N, device=dev, generator=g)
    wt = w.t().contiguous()
    out, gx, gw = (torch.empty(M, N, device=dev), torch.empty(M, K, device=dev),
                   torch.empty(N, K, device=dev))
    row = []
    for name, own, lib in (
            ('fwd', lambda: _native.linear_forward(x, w, None, out=out),
             lambda: torch.mm(x, w.t(), out=out)),
            ('dgrad', lambda: _native.linear_dgrad(go, wt, out=gx), lambda: torch.mm(go, w, out=gx)),
            ('wgrad', lambda: _native.linear_wgrad(go, x, out=gw),
             lambda: torch.mm(go.t(), x, out=gw))):
        ts = []
        for mode in ('fp32', 'split'):
            prev = _native.set_gemm_mode(mode)
            ts.append(timeit(own))
            _native.set_gemm_mode(prev)
        row.append(f'{name} {ts[0]:7.1f} / {ts[1]:7.1f} / {timeit(lib):7.1f}')
    print(f'({M:6d}, {K:5d}, {N:4d}): ' + ' | '.join(row), flush=True)

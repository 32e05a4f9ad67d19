"""Times the one-kernel SAGE layer forward (csrc/sage_fused.hip) against the two-launch schedule
(SpMM + own GEMM) at the ogbn-products shape.  Usage: python scripts/fused_probe.py [--scale s]"""
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
args = ap.parse_args()
dev = torch.device('cuda:0')
x0, _, ei, _ = products_like(seed=1, scale=args.scale)
N = x0.size(0)
h = pga.EdgeIndex(ei.to(dev), (N, N))
fwd = h.by_dst()
fwd.hub


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device=dev).manual_seed(0)
for F, Fo in ((256, 256), (100, 256)):
    buf = torch.empty(N, 2 * F, device=dev)
    buf[:, F:] = torch.randn(N, F, device=dev, generator=g)
    xsrc = buf[:, F:].contiguous() if F == 100 else buf[:, F:]
    w = torch.randn(Fo, 2 * F, device=dev, generator=g) * 0.05
    b = torch.randn(Fo, device=dev, generator=g)
    out = torch.empty(N, Fo, device=dev)
    ref = torch.empty(N, Fo, device=dev)

    def two():
        _native.spmm_csr(fwd.ptr, fwd.idx, xsrc, 'mean', n_rows=N, hub=fwd.hub, out=buf[:, :F])
        _native.linear_forward(buf, w, b, relu=True, out=ref)

    def one():
        _native.sage_layer_forward(fwd.ptr, fwd.idx, xsrc, buf[:, F:], w, b, 'mean', True,
                                   buf[:, :F], out, hub=fwd.hub, save_agg=True)

    t_spmm = timeit(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, xsrc, 'mean', n_rows=N,
                                             hub=fwd.hub, out=buf[:, :F]))
    t2, t1 = timeit(two), timeit(one)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'F={F} Fo={Fo}: SpMM {t_spmm:.3f} ms, SpMM+GEMM {t2:.3f} ms, one kernel {t1:.3f} ms '
          f'(max rel diff {err:.2e})', flush=True)

"""Probe: does the F = 256 SpMM get faster when run as column slabs whose source table fits the
256 MB Infinity Cache?  Times the existing kernel on strided / contiguous slabs of the
products-shaped problem.  Informational."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import EdgeIndex, _native  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402

dev = torch.device('cuda:0')
x, y, ei, _ = products_like(seed=1, scale=1.0)
N = x.size(0)
g = EdgeIndex(ei.to(dev), (N, N))
fwd = g.by_dst()
h = torch.randn(N, 256, device=dev)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = torch.empty(N, 256, device=dev)
t = timeit(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, h, 'mean', n_rows=N, hub=fwd.hub, out=out))
print(f'full F=256                     : {t:7.3f} ms')
for W in (16, 32, 64, 128):
    S = 256 // W
    def strided():
        for s in range(S):
            _native.spmm_csr(fwd.ptr, fwd.idx, h[:, s * W:(s + 1) * W], 'mean', n_rows=N,
                             hub=fwd.hub, out=out[:, s * W:(s + 1) * W])
    t = timeit(strided, 3)
    hc = h[:, :W].contiguous()
    oc = torch.empty(N, W, device=dev)
    tc = timeit(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, hc, 'mean', n_rows=N, hub=fwd.hub,
                                         out=oc), 3)
    print(f'{S:2d} slabs of {W:3d} (strided views) : {t:7.3f} ms   | one contiguous [N,{W}] slab: '
          f'{tc:6.3f} ms x {S} = {tc * S:7.3f} ms')

# ---- second probe: the bench's actual launches ---------------------------------------------------
bwd = g.by_src()
scale = fwd.inv_degree()
gcat = torch.randn(N, 512, device=dev)


def run_slabs(ptr, idx, src, dst, widths, hub, **kw):
    c = 0
    for W in widths:
        _native.spmm_csr(ptr, idx, src[:, c:c + W], kw.get('reduce', 'sum'), n_rows=N, hub=hub,
                         out=dst[:, c:c + W], src_scale=kw.get('scale'),
                         accumulate=kw.get('acc', False))
        c += W


for widths in ([256], [128, 128], [64] * 4, [96, 96, 64], [48] * 5 + [16], [32] * 8):
    t = timeit(lambda: run_slabs(bwd.ptr, bwd.idx, gcat[:, :256], gcat[:, 256:], widths, bwd.hub,
                                 scale=scale, acc=True), 3)
    print(f'backward (scaled, accumulate, ld 512) slabs {widths}: {t:7.3f} ms')
x100 = torch.randn(N, 100, device=dev)
buf = torch.empty(N, 200, device=dev)
for widths in ([100], [52, 48], [64, 36], [36, 32, 32], [48, 52]):
    t = timeit(lambda: run_slabs(fwd.ptr, fwd.idx, x100, buf, widths, fwd.hub, reduce='mean'), 3)
    print(f'layer-0 forward F=100 slabs {widths}: {t:7.3f} ms')
y = torch.randn(N, 96, device=dev)
for widths in ([48], [24, 24], [32, 16]):
    t = timeit(lambda: run_slabs(fwd.ptr, fwd.idx, y[:, :48], y[:, 48:], widths, fwd.hub,
                                 reduce='mean', acc=True), 3)
    print(f'layer-2 pre-mode F=48 (accumulate) slabs {widths}: {t:7.3f} ms')

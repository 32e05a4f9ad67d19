"""Informational: what a row gather costs as a function of the 128-byte lines it touches per row,
at a fixed row pitch (the products graph, sum aggregation).  The first W floats of every source row
are read from rows pitched at 1024 / 1152 bytes: the time a gather of losslessly COMPRESSED
256-float rows (zeros of the ReLU removed: a 32-byte bit mask + ~128 packed values = 4.4 lines
instead of 8) could reach."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import EdgeIndex, _native  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402

dev = torch.device('cuda:0')
x, y, ei, _ = products_like(seed=1, scale=float(os.environ.get('SCALE', 1.0)))
N, E = x.size(0), ei.size(1)
g = EdgeIndex(ei.to(dev), (N, N)).by_dst()


def timeit(fn, n=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for pitch in (256, 288):
    buf = torch.randn(N, pitch, device=dev)
    for W in (256, 192, 160, 144, 128, 96, 64):
        src = buf[:, :W]
        out = torch.empty(N, W, device=dev)
        t = timeit(lambda: _native.spmm_csr(g.ptr, g.idx, src, 'sum', n_rows=N, hub=g.hub,
                                            out=out))
        lines = (W * 4 + 127) // 128
        print(f'pitch {pitch * 4:5d} B, first {W:3d} floats ({lines} lines / row): {t:6.2f} ms '
              f'({E * lines * 128 / t / 1e9:5.2f} TB/s of lines, {E / t / 1e6:6.2f} G rows/s)')

# the same aggregation from COMPRESSED rows (pygamd_rows_compress): a ReLU-like block, half zeros
for density in (0.5, 0.25):
    h = torch.randn(N, 256, device=dev)
    h[torch.rand(N, 256, device=dev) >= density] = 0
    z = _native.rows_compress(h)
    out = torch.empty(N, 256, device=dev)
    td = timeit(lambda: _native.spmm_csr(g.ptr, g.idx, h, 'sum', n_rows=N, hub=g.hub, out=out))
    tz = timeit(lambda: _native.spmm_csr(g.ptr, g.idx, z, 'sum', n_rows=N, hub=g.hub, out=out,
                                         compressed_width=256))
    tc = timeit(lambda: _native.rows_compress(h, out=z))
    lines = (32 + 4 * 256 * density) / 128
    print(f'stand-alone aggregation, F = 256, {density:.0%} non-zero: dense rows {td:6.2f} ms, '
          f'compressed rows {tz:6.2f} ms ({lines:.1f} lines / row on average), compressing the '
          f'block {tc:5.2f} ms')
    del h, z, out

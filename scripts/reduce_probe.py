"""Informational: time every reduction of the fused aggregation (and the unfused scatter path) at
the products shape, F = 256, forward and backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import EdgeIndex  # noqa: E402
from pytorch_geometric_amd._functions import SpmmFunction  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402
from pytorch_geometric_amd import utils as U  # noqa: E402

dev = torch.device('cuda:0')
F = int(os.environ.get('F', 256))
x, y, ei, _ = products_like(seed=1, scale=float(os.environ.get('SCALE', 1.0)))
N, E = x.size(0), ei.size(1)
ei = ei.to(dev)
g = EdgeIndex(ei, (N, N))
g.by_dst(); g.by_src()
h = torch.randn(N, F, device=dev, requires_grad=True)
go = torch.randn(N, F, device=dev)


def timeit(fn, n=3):
    fn()
    fn()   # (the first call of a shape also grows the caching allocator)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


GB = E * (4 * F + 8) / 1e9
for red in ('sum', 'mean', 'max', 'min'):
    out = None

    def fwd():
        global out
        out = SpmmFunction.apply(h, None, g, red, 'coo')

    def bwd():
        h.grad = None
        out.backward(go, retain_graph=True)

    tf = timeit(fwd)
    tb = timeit(bwd)
    print(f'fused {red:4s}: fwd {tf:7.2f} ms ({GB / tf:5.2f} TB/s algorithmic)   bwd {tb:7.2f} ms')
if os.environ.get('UNFUSED'):
    GBu = E * 4 * F / 1e9
    for red in ('sum', 'mean', 'max', 'mul'):
        def fwd():
            global out
            out = U.scatter(h[ei[0]], ei[1], 0, N, red)

        def bwd():
            h.grad = None
            out.backward(go, retain_graph=True)

        tf = timeit(fwd, 2)
        tb = timeit(bwd, 2)
        print(f'unfused index_select + scatter {red:4s}: fwd {tf:7.2f} ms   bwd {tb:7.2f} ms   '
              f'([E, F] messages = {GBu:.1f} GB)')

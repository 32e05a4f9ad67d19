"""Informational timings (ms per fwd+bwd step) of BASELINE.json configs 1, 3, 5 on one MI355X.
Not bench lines — bench.py reports config 2, the configuration the metric is quoted on."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.nn import GAT, GCN, RGCNConv  # noqa: E402

dev = torch.device('cuda:0')
if not os.environ.get('PYTORCH_TUNABLEOP_TUNING') and not os.environ.get('NO_TUNED_GEMM'):
    from pytorch_geometric_amd.tuning import CSV, enable_tuned_gemms
    print('tuned GEMM table:', enable_tuned_gemms(os.environ.get('TUNED_GEMM', CSV)))


def timeit(fn, warm=3, steps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


g = torch.Generator().manual_seed(0)
n, pairs = 2708, 5278
u, v = torch.randint(0, n, (pairs, ), generator=g), torch.randint(0, n, (pairs, ), generator=g)
ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])]).to(dev)
x = torch.rand(n, 1433, generator=g).to(dev)
model = GCN(1433, 16, num_layers=2, out_channels=7, cached=True).to(dev)


def step1():
    model.zero_grad()
    model(x, ei).sum().backward()


print(f'config1 GCN/Cora-shape      : {timeit(step1):8.3f} ms/step')
# the same step captured into a HIP graph (launch-bound: ~40 small kernels)
for p_ in model.parameters():
    p_.grad = torch.zeros_like(p_)


def step1g():
    model.zero_grad(set_to_none=False)
    model(x, ei).sum().backward()


if not os.environ.get('SKIP_GRAPH'):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step1g()
    torch.cuda.current_stream().wait_stream(side)
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        step1g()
    print(f'config1 as one hipGraph     : {timeit(cg.replay, warm=5, steps=50):8.3f} ms/step')

n, e = 169_343, 1_166_243
ei = torch.randint(0, n, (2, e), generator=g).to(dev)
x = torch.randn(n, 128, generator=g).to(dev)
model3 = GAT(128, 256, num_layers=3, out_channels=40, heads=8).to(dev)


def step3():
    model3.zero_grad()
    model3(x, ei).sum().backward()


t = timeit(step3)
print(f'config3 GAT/arxiv-shape     : {t:8.3f} ms/step  ({3 * (e + n) / t / 1e3:.1f} M edges/s)')

n, e, R = 14_541, 544_230, 474
ei = torch.randint(0, n, (2, e), generator=g).to(dev)
et = (torch.rand(e, generator=g).pow(4) * R).long().clamp(max=R - 1).to(dev)
emb = torch.nn.Parameter(torch.randn(n, 500, device=dev))
c1 = RGCNConv(500, 500, R, num_blocks=5).to(dev)
c2 = RGCNConv(500, 500, R, num_blocks=5).to(dev)


def step5():
    c1.zero_grad(); c2.zero_grad(); emb.grad = None
    c2(c1(emb, ei, et).relu(), ei, et).sum().backward()


t = timeit(step5, warm=2, steps=5)
print(f'config5 RGCN/FB15k-237-shape: {t:8.3f} ms/step  ({2 * e / t / 1e3:.1f} M edges/s)')

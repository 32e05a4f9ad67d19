"""Where the host time of an eager (un-captured) config-1 step goes: cProfile of 200 steps of the
2-layer GCN on the Cora shape (launch-bound: the GPU work of a step is ~0.12 ms).
Usage: python scripts/eager_probe.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.nn import GCN  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
n, pairs = 2708, 5278
u, v = torch.randint(0, n, (pairs, ), generator=g), torch.randint(0, n, (pairs, ), generator=g)
ei = torch.stack([torch.cat([u, v]), torch.cat([v, u])]).to(dev)
x = torch.rand(n, 1433, generator=g).to(dev)
model = GCN(1433, 16, num_layers=2, out_channels=7, cached=True).to(dev)


def step():
    model.zero_grad()
    model(x, ei).sum().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    print(f'eager: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)

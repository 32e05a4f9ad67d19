"""Informational: BASELINE config 3 alone (GAT(128, 256, num_layers=3, out_channels=40, heads=8) on
the ogbn-arxiv shape), for profiling: `rocprofv3 --kernel-trace --stats -- python scripts/time_gat.py`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.nn import GAT  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
n, e = 169_343, 1_166_243
ei = torch.randint(0, n, (2, e), generator=g).to(dev)
x = torch.randn(n, 128, generator=g).to(dev)
model = GAT(128, 256, num_layers=3, out_channels=40, heads=8).to(dev)


def step():
    model.zero_grad()
    model(x, ei).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = int(os.environ.get('STEPS', 10))
for _ in range(steps):
    step()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps * 1e3
print(f'config3 GAT/arxiv-shape: {t:8.3f} ms/step  ({3 * (e + n) / t / 1e3:.1f} M edges/s)')

"""Informational: BASELINE config 5 alone (RGCNConv(500, 500, 474, num_blocks=5) x 2 on the
FB15k-237 shape), for profiling: `rocprofv3 --kernel-trace --stats -- python scripts/time_rgcn.py`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.nn import RGCNConv  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
n, e, R = 14_541, 544_230, 474
ei = torch.randint(0, n, (2, e), generator=g).to(dev)
et = (torch.rand(e, generator=g).pow(4) * R).long().clamp(max=R - 1).to(dev)
emb = torch.nn.Parameter(torch.randn(n, 500, device=dev))
c1 = RGCNConv(500, 500, R, num_blocks=5).to(dev)
c2 = RGCNConv(500, 500, R, num_blocks=5).to(dev)


def step():
    c1.zero_grad(); c2.zero_grad(); emb.grad = None
    c2(c1(emb, ei, et).relu(), ei, et).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = int(os.environ.get('STEPS', 10))
for _ in range(steps):
    step()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps * 1e3
print(f'config5 RGCN/FB15k-237-shape: {t:8.3f} ms/step  ({2 * e / t / 1e3:.1f} M edges/s), '
      f'pairs S = {c1._handle_cache[-1].num_pairs}')

"""A model built from SAGEConv LAYERS (what most PyG scripts do — not the `GraphSAGE` class) on
the bench's workload: 3 layers 100 -> 256 -> 256 -> 47 with ReLU between them, ogbn-products shape,
CE on the 8 % split + Adam.  Per step: the one-kernel layer as one autograd node per conv
(default) against propagate + Linear (PYGAMD_SAGE_LAYER_NODE=0), and the `GraphSAGE` class (the
fused whole-stack schedule) for reference.
Usage: python scripts/time_layered_sage.py [--scale 1.0] [--steps 10]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.datasets import products_like  # noqa: E402
from pytorch_geometric_amd.nn import GraphSAGE, SAGEConv  # noqa: E402
from pytorch_geometric_amd.nn.models import _fused_sage  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--steps', type=int, default=10)
args = ap.parse_args()
dev = torch.device('cuda:0')
x, y, ei, C = products_like(seed=1, scale=args.scale, skewed=True, dtype=torch.int64)
N, E = x.size(0), ei.size(1)
x, y, ei = x.to(dev), y.to(dev), ei.to(dev)
train_idx = torch.randperm(N, generator=torch.Generator().manual_seed(7))[:int(0.0803 * N)].to(dev)
y_train = y[train_idx]


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.convs = torch.nn.ModuleList([SAGEConv(100, 256), SAGEConv(256, 256),
                                          SAGEConv(256, C)])

    def forward(self, x, edge_index):
        for conv in self.convs[:-1]:
            x = F.relu(conv(x, edge_index))
        return self.convs[-1](x, edge_index)


def timed(model, steps):
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(x, ei)[train_idx], y_train)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(loss)


torch.manual_seed(0)
net = Net().to(dev)
state = {k: v.clone() for k, v in net.state_dict().items()}
for flag, name in ((True, 'SAGEConv layers, one-kernel layer nodes'),
                   (False, 'SAGEConv layers, propagate + Linear   ')):
    _fused_sage.LAYER_NODE = flag
    net.load_state_dict(state)
    ms, loss = timed(net, args.steps)
    print(f'{name}: {ms:8.2f} ms/step  ({3 * E / ms / 1e6:.2f} G edges/s)  loss {loss:.5f}')
_fused_sage.LAYER_NODE = True
torch.manual_seed(0)
model = GraphSAGE(100, 256, num_layers=3, out_channels=C).to(dev)
ms, loss = timed(model, args.steps)
print(f'GraphSAGE class, fused whole-stack schedule : {ms:8.2f} ms/step  '
      f'({3 * E / ms / 1e6:.2f} G edges/s)  loss {loss:.5f}')

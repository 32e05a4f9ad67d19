"""A 3-layer GCN (100 -> 256 -> 256 -> 47, GCNConv layers + ReLU) at the ogbn-products shape, CE on
the 8 % split + Adam: GCNConv aggregating at the narrower width (default: the first layer runs
lin(propagate(x)) and its backward needs no aggregation at all) against the reference's order
propagate(lin(x)) everywhere (`aggregate_first = False`).
Usage: python scripts/time_layered_gcn.py [--scale 1.0] [--steps 10]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.datasets import products_like  # noqa: E402
from pytorch_geometric_amd.nn import GCNConv  # noqa: E402

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
        self.convs = torch.nn.ModuleList([GCNConv(100, 256, cached=True),
                                          GCNConv(256, 256, cached=True),
                                          GCNConv(256, C, cached=True)])

    def forward(self, x, edge_index):
        for conv in self.convs[:-1]:
            x = F.relu(conv(x, edge_index))
        return self.convs[-1](x, edge_index)


torch.manual_seed(0)
net = Net().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    loss = F.cross_entropy(net(x, ei)[train_idx], y_train)
    loss.backward()
    opt.step()
    return loss


for first, name in ((True, 'aggregate at the narrower width'), (False, "the reference's order")):
    for conv in net.convs:
        conv.aggregate_first = first
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(f'GCNConv x 3, {name:32s}: {ms:8.2f} ms/step  ({3 * (E + N) / ms / 1e6:.2f} G edges/s)')

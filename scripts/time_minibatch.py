"""Informational: BASELINE config 4 (GraphSAGE 3-layer + NeighborLoader [15,10,5], batch 1024) on a
synthetic papers100M-shaped graph scaled to `--scale` (default 1/16: 6.9 M nodes, 101 M edges,
F = 128, 172 classes), one MI355X.  Prints per-batch times of sampling, feature gather and the
training step (with trim_to_layer) and the aggregated edges/s of the step."""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.datasets import powerlaw_undirected  # noqa: E402
from pytorch_geometric_amd.loader import NeighborLoader  # noqa: E402
from pytorch_geometric_amd.nn import GraphSAGE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1 / 16)
ap.add_argument('--batches', type=int, default=30)
args = ap.parse_args()
dev = torch.device('cuda:0')
N = int(111_059_956 * args.scale)
E = int(1_615_685_872 * args.scale) // 2 * 2
t0 = time.perf_counter()
ei = powerlaw_undirected(N, E, seed=3).to(dev)
x = torch.randn(N, 128, device=dev)
y = torch.randint(0, 172, (N, ), device=dev)
print(f'graph N={N} E={E} built in {time.perf_counter() - t0:.1f}s')
model = GraphSAGE(128, 256, num_layers=3, out_channels=172).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
loader = NeighborLoader(x, ei, [15, 10, 5], batch_size=1024, y=y, shuffle=True,
                        input_nodes=torch.arange(N // 10, device=dev))
t_s = t_g = t_t = 0.0
edges = 0
it = iter(loader)
for b in range(args.batches + 5):
    seeds = loader.input_nodes[torch.randint(0, loader.input_nodes.numel(), (1024, ), device=dev)]
    seeds = torch.unique(seeds)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = loader.sampler.sample_from_nodes(seeds)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    from pytorch_geometric_amd import _native
    xb = _native.gather_rows(x, out.node)
    yb = y[out.node[:seeds.numel()]]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.zero_grad()
    from pytorch_geometric_amd import EdgeIndex
    graph = EdgeIndex.from_sorted_batch(torch.stack([out.row, out.col]), out.node.numel(),
                                        max_in_degree=15)
    logits = model(xb, graph,
                   num_sampled_nodes_per_hop=out.num_sampled_nodes,
                   num_sampled_edges_per_hop=out.num_sampled_edges)[:seeds.numel()]
    F.cross_entropy(logits, yb).backward()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if b >= 5:
        t_s += t1 - t0; t_g += t2 - t1; t_t += t3 - t2
        ne = out.num_sampled_edges
        edges += ne[0] + (ne[0] + ne[1]) + sum(ne)  # edges each trimmed layer aggregates
n = args.batches
print(f'per batch: sample {t_s / n * 1e3:.2f} ms, gather {t_g / n * 1e3:.2f} ms, '
      f'train step {t_t / n * 1e3:.2f} ms; nodes/batch {out.node.numel()}, '
      f'edges/batch {out.row.numel()}')
print(f'aggregated edges/s (trimmed, incl. sampling+gather): {edges / (t_s + t_g + t_t) / 1e6:.1f} M')

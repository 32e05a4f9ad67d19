"""Config 4's batches through a model written as SAGEConv LAYERS on the whole sampled subgraph (the
style of examples/multi_gpu/distributed_sampling.py:20-40; no trim_to_layer): propagate + Linear
per layer with the atomic backward of single-use batch handles (what such a model gets today)
against the one-kernel layer node, which needs the by-source sort of the batch (once per batch,
shared by the three layers).  Usage: python scripts/time_minibatch_layers.py [--scale 0.0625]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import EdgeIndex, _native  # noqa: E402
from pytorch_geometric_amd.datasets import powerlaw_undirected  # noqa: E402
from pytorch_geometric_amd.loader import NeighborLoader  # noqa: E402
from pytorch_geometric_amd.nn import SAGEConv  # noqa: E402
from pytorch_geometric_amd.nn.models import _fused_sage  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1 / 16)
ap.add_argument('--batches', type=int, default=30)
args = ap.parse_args()
dev = torch.device('cuda:0')
N = int(111_059_956 * args.scale)
E = int(1_615_685_872 * args.scale) // 2 * 2
ei = powerlaw_undirected(N, E, seed=3).to(dev)
x = torch.randn(N, 128, device=dev)
y = torch.randint(0, 172, (N, ), device=dev)
loader = NeighborLoader(x, ei, [15, 10, 5], batch_size=1024, y=y, shuffle=True,
                        input_nodes=torch.arange(N // 10, device=dev))


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.convs = torch.nn.ModuleList([SAGEConv(128, 256), SAGEConv(256, 256),
                                          SAGEConv(256, 172)])

    def forward(self, x, graph):
        for conv in self.convs[:-1]:
            x = F.relu(conv(x, graph))
        return self.convs[-1](x, graph)


torch.manual_seed(0)
model = Net().to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
real_ok = _fused_sage._graph_ok


def any_handle_ok(edge_index, n):
    if isinstance(edge_index, EdgeIndex):
        return edge_index.sparse_size == (n, n)
    return real_ok(edge_index, n)


g = torch.Generator(device=dev).manual_seed(1)
for name, ok in (('propagate + Linear, atomic backward', real_ok),
                 ('one-kernel layer nodes (+ by-source sort)', any_handle_ok)):
    _fused_sage._graph_ok = ok
    t_t, edges = 0.0, 0
    for b in range(args.batches + 5):
        seeds = torch.unique(loader.input_nodes[torch.randint(
            0, loader.input_nodes.numel(), (1024, ), device=dev, generator=g)])
        out = loader.sampler.sample_from_nodes(seeds)
        xb = _native.gather_rows(x, out.node)
        yb = y[out.node[:seeds.numel()]]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        graph = EdgeIndex.from_sorted_batch(torch.stack([out.row, out.col]), out.node.numel(),
                                            max_in_degree=15)
        opt.zero_grad()
        logits = model(xb, graph)[:seeds.numel()]
        F.cross_entropy(logits, yb).backward()
        opt.step()
        torch.cuda.synchronize()
        if b >= 5:
            t_t += time.perf_counter() - t0
            edges += 3 * out.row.numel()
    print(f'{name:44s}: {t_t / args.batches * 1e3:7.3f} ms per batch step  '
          f'({edges / t_t / 1e6:.0f} M aggregated edges/s), {out.node.numel()} nodes / '
          f'{out.row.numel()} edges in the last batch')
_fused_sage._graph_ok = real_ok

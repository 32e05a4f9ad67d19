"""Where do the parameter-gradient differences of the products step come from?  At 1/16 of the
shape: the fp32 oracle (CPU) and every GPU schedule against an fp64 evaluation of the same model.
Usage: python scripts/parity_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyg_oracle as O  # noqa: E402
from pytorch_geometric_amd import _native  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402
from pytorch_geometric_amd.nn import GraphSAGE  # noqa: E402
from pytorch_geometric_amd.nn.models import _fused_sage  # noqa: E402

dev = torch.device('cuda:0')
x, y, ei, c = products_like(seed=1, scale=1 / 16)
N = x.size(0)
ti = torch.randperm(N, generator=torch.Generator().manual_seed(7))[:int(0.0803 * N)]
torch.manual_seed(0)
model = GraphSAGE(100, 256, num_layers=3, out_channels=c)
st = {k: v.clone() for k, v in model.state_dict().items()}
names = ('lin_l.weight', 'lin_l.bias', 'lin_r.weight')


def cpu(dtype):
    params = [tuple(st[f'convs.{i}.{n}'].to(dtype).requires_grad_(True) for n in names)
              for i in range(3)]
    out = O.graphsage(x.to(dtype), ei, params)
    loss = F.cross_entropy(out[ti], y[ti])
    loss.backward()
    return out.detach(), {f'convs.{i}.{n}': params[i][k].grad for i in range(3)
                          for k, n in enumerate(names)}


out64, g64 = cpu(torch.float64)
out32, g32 = cpu(torch.float32)


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())


print('CPU fp32 oracle vs fp64: out %.2e' % rel(out32, out64),
      {k: '%.1e' % rel(g32[k], g64[k]) for k in g64})
model = model.to(dev)
for variant, fuse_bwd in ((1, False), (2, False), (2, True)):
    _native.SAGE_FUSED_VARIANT = variant
    _fused_sage.FUSE_BWD = fuse_bwd
    model.zero_grad()
    out = model(x.to(dev), ei.to(dev))
    loss = F.cross_entropy(out[ti.to(dev)], y.to(dev)[ti.to(dev)])
    loss.backward()
    got = {k: p.grad for k, p in model.named_parameters()}
    print(f'GPU variant={variant} fuse_bwd={fuse_bwd} vs fp64: out %.2e' % rel(out.detach(), out64),
          {k: '%.1e' % rel(got[k], g64[k]) for k in g64})
    print('   vs fp32 oracle:', {k: '%.1e' % rel(got[k], g32[k].double()) for k in g64})

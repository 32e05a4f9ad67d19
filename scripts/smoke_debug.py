"""smoke()'s comparison with every tensor's error printed (relative to the tensor's largest
magnitude), for bisecting with PYGAMD_BINDING / PYGAMD_FUSE_BWD / PYGAMD_FUSE_LAYER /
PYGAMD_FUSED_VARIANT; also prints how close the nearest hidden pre-activation of the fp64 oracle
is to the ReLU kink (below ~1e-6 the gradient comparison is ill-conditioned: the activation lands
on the other side of 0 in fp32 and a whole row of the next weight gradient moves).
Usage: python scripts/smoke_debug.py [scale_denominator] [xgrad 0|1] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyg_oracle as O  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402
from pytorch_geometric_amd.nn import GraphSAGE  # noqa: E402

den = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xgrad = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device('cuda:0')
x, y, ei, c = products_like(seed=seed, scale=1 / den)
torch.manual_seed(0)
model = GraphSAGE(100, 256, num_layers=3, out_channels=c)
st = {k: v.clone() for k, v in model.state_dict().items()}
params = [tuple(st[f'convs.{i}.{n}'].double().requires_grad_(True)
                for n in ('lin_l.weight', 'lin_l.bias', 'lin_r.weight')) for i in range(3)]
xr = x.double().requires_grad_(True)
ref = O.graphsage(xr, ei, params)
torch.nn.functional.cross_entropy(ref, y).backward()
with torch.no_grad():  # distance of the nearest hidden pre-activation to the ReLU kink (fp64)
    h, kink = x.double(), float('inf')
    for wl, b, wr in params[:-1]:
        pre = O.sage_conv(h, ei, wl, b, wr, 'mean')
        kink = min(kink, float(pre.abs().min()))
        h = pre.relu()
model = model.to(dev)
xg = x.to(dev).requires_grad_(xgrad)
out = model(xg, ei.to(dev))
torch.nn.functional.cross_entropy(out, y.to(dev)).backward()
torch.cuda.synchronize()


def rel(a, b):
    return float((a.detach().cpu().double() - b).abs().max() / b.abs().max().clamp(min=1e-30))


env = {k: os.environ.get(k) for k in ('PYGAMD_BINDING', 'PYGAMD_FUSE_BWD', 'PYGAMD_FUSE_LAYER',
                                      'PYGAMD_FUSED_VARIANT') if os.environ.get(k)}
print(f'N={x.size(0)} E={ei.size(1)} seed={seed} xgrad={xgrad} env={env}; nearest hidden '
      f'pre-activation to 0 in the fp64 oracle: {kink:.2e}')
print('  out', f'{rel(out, ref.detach()):.2e}', 'grad_x', f'{rel(xg.grad, xr.grad):.2e}' if xgrad else '-')
for i, conv in enumerate(model.convs):
    print(f'  convs.{i}:', ' '.join(
        f'{n} {rel(g, w.grad):.2e}' for n, g, w in (('W_l', conv.lin_l.weight.grad, params[i][0]),
                                                     ('b', conv.lin_l.bias.grad, params[i][1]),
                                                     ('W_r', conv.lin_r.weight.grad, params[i][2]))))

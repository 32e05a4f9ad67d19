"""Which ATen copies / fills / adds remain in a config-5 (RGCN) and a config-3 (GAT) step, with
their shapes and the Python line that issued them (torch.profiler, record_shapes + with_stack).
Usage: python scripts/copy_probe.py [rgcn|gat]"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd.nn import GAT, RGCNConv  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'rgcn'

if which == 'rgcn':
    n, e, R = 14_541, 544_230, 474
    ei = torch.randint(0, n, (2, e), generator=g).to(dev)
    et = (torch.rand(e, generator=g).pow(4) * R).long().clamp(max=R - 1).to(dev)
    emb = torch.nn.Parameter(torch.randn(n, 500, device=dev))
    c1 = RGCNConv(500, 500, R, num_blocks=5).to(dev)
    c2 = RGCNConv(500, 500, R, num_blocks=5).to(dev)

    def step():
        c1.zero_grad(); c2.zero_grad(); emb.grad = None
        c2(c1(emb, ei, et).relu(), ei, et).sum().backward()
else:
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
             with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

WATCH = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::clone',
         'aten::contiguous', 'aten::mul', 'aten::sum', 'aten::cat', 'aten::index', 'aten::relu',
         'aten::threshold_backward', 'aten::expand')
rows = Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    here = [s for s in (ev.stack or []) if 'pytorch_geometric_amd' in s or 'scripts/' in s]
    where = here[0].split('pytorch_geometric_amd/')[-1] if here else '(autograd engine / C++)'
    dev_us = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
    rows[(ev.name, str(ev.input_shapes)[:70], where[:80])] += 1
    rows[('~us', ev.name, str(ev.input_shapes)[:70], where[:80])] += dev_us
print(f'{which}: ATen glue in one step')
for key, cnt in sorted((k, v) for k, v in rows.items() if k[0] != '~us'):
    us = rows[('~us', ) + key]
    print(f'{cnt:3d} x {key[0]:26s} {us:8.1f} us  {key[1]:70s}  {key[2]}')

"""Generates tests/golden/golden_aggr_v1.pt by running the REAL reference (PyG, /root/reference)
on CPU: SoftmaxAggregation / PowerMeanAggregation (nn/aggr/basic.py:142-296).  Build container
only:   PYTHONPATH=/root/reference python tests/golden/make_golden_aggr.py
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get('PYG_REFERENCE', '/root/reference'))
import torch_geometric  # noqa: E402
from torch_geometric.nn.aggr import PowerMeanAggregation, SoftmaxAggregation  # noqa: E402


def gen(seed):
    return torch.Generator().manual_seed(seed)


g = gen(60)
x = torch.randn(80, 6, generator=g)
index = torch.randint(0, 12, (80, ), generator=g)
index[index == 7] = 3                      # group 7 empty
index, _ = index.sort()
ptr = torch.zeros(15, dtype=torch.long)
ptr[1:] = torch.bincount(index, minlength=14).cumsum(0)
G = {'meta': {'torch': torch.__version__, 'pyg': torch_geometric.__version__},
     'x': x, 'index': index, 'ptr': ptr, 'dim_size': 14}


def case(aggr, seed, use_ptr=False, positive=False):
    xx = (x.abs() + 0.1 if positive else x).clone().requires_grad_(True)
    kw = dict(ptr=ptr) if use_ptr else dict(index=index, dim_size=14)
    out = aggr(xx, **kw)
    go = torch.randn(out.shape, generator=gen(seed))
    params = list(aggr.parameters())
    grads = torch.autograd.grad(out, [xx] + params, go)
    return {'out': out.detach(), 'grad_out': go, 'grad_x': grads[0],
            'grad_param': grads[1].detach() if params else None}


C = {}
C['softmax_t1'] = case(SoftmaxAggregation(), 600)
C['softmax_t05_ptr'] = case(SoftmaxAggregation(t=0.5), 601, use_ptr=True)
C['softmax_semi'] = case(SoftmaxAggregation(t=2.0, semi_grad=True), 602)
C['softmax_learn'] = case(SoftmaxAggregation(t=0.7, learn=True), 603)
C['softmax_learn_channels'] = case(SoftmaxAggregation(t=1.3, learn=True, channels=6), 604)
C['powermean_p1'] = case(PowerMeanAggregation(), 605)
C['powermean_p2'] = case(PowerMeanAggregation(p=2.0), 606, positive=True)
C['powermean_p3_ptr_clamped'] = case(PowerMeanAggregation(p=3.0), 607, use_ptr=True)
C['powermean_learn_channels'] = case(PowerMeanAggregation(p=1.5, learn=True, channels=6), 608,
                                     positive=True)
G['cases'] = C
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_aggr_v1.pt')
torch.save(G, out)
print('wrote', out, os.path.getsize(out), 'bytes')

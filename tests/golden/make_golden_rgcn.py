"""Generates tests/golden/golden_rgcn_v1.pt by running the REAL reference (PyG, /root/reference)
on CPU: FastRGCNConv and the node-index ("featureless") inputs of RGCNConv / FastRGCNConv
(nn/conv/rgcn_conv.py:164-374), on the graph of golden_v1.pt.  Build container only:

    PYTHONPATH=/root/reference python tests/golden/make_golden_rgcn.py
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get('PYG_REFERENCE', '/root/reference'))
import torch_geometric  # noqa: E402
from torch_geometric.nn import FastRGCNConv, RGCNConv  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
base = torch.load(os.path.join(HERE, 'golden_v1.pt'), map_location='cpu', weights_only=False)
gr = base['graph']
x16, edge_index, edge_type = gr['x'], gr['edge_index'], gr['edge_type']
N = x16.size(0)
G = {'meta': {'torch': torch.__version__, 'pyg': torch_geometric.__version__}}


def gen(seed):
    return torch.Generator().manual_seed(seed)


def case(conv, seed, x):
    conv.eval()
    xx = x.clone().requires_grad_(True) if (x is not None and x.is_floating_point()) else x
    out = conv(xx, edge_index, edge_type)
    go = torch.randn(out.shape, generator=gen(seed))
    params = list(conv.parameters())
    leaves = ([xx] if isinstance(xx, torch.Tensor) and xx.requires_grad else []) + params
    grads = torch.autograd.grad(out, leaves, go, allow_unused=True)
    gx = grads[0] if len(leaves) > len(params) else None
    gp = grads[len(leaves) - len(params):]
    return {'state': {k: v.detach().clone() for k, v in conv.state_dict().items()},
            'out': out.detach(), 'grad_out': go, 'grad_x': gx,
            'grad_params': {n: (None if g_ is None else g_.detach())
                            for (n, _), g_ in zip(conv.named_parameters(), gp)}}


torch.manual_seed(30)
x_idx = torch.randint(0, 16, (N, ), generator=gen(31))
G['x_idx'] = x_idx
L = {}
L['fast'] = case(FastRGCNConv(16, 10, num_relations=5), 400, x16)
L['fast_add'] = case(FastRGCNConv(16, 10, num_relations=5, aggr='add'), 401, x16)
L['fast_blocks'] = case(FastRGCNConv(16, 12, num_relations=5, num_blocks=4), 402, x16)
L['fast_bases'] = case(FastRGCNConv(16, 10, num_relations=5, num_bases=3), 403, x16)
# (FastRGCNConv looks node-index inputs up by the SOURCE NODE id, rgcn_conv.py:357-359, so it is
# only meaningful with x = None, in_channels = num_nodes: the `fast_none` case below)
L['rgcn_index'] = case(RGCNConv(16, 10, num_relations=5), 406, x_idx)
L['rgcn_index_max'] = case(RGCNConv(16, 10, num_relations=5, aggr='max'), 407, x_idx)
L['rgcn_none'] = case(RGCNConv(N, 10, num_relations=5), 408, None)      # x = None -> arange(N)
L['fast_none'] = case(FastRGCNConv(N, 10, num_relations=5), 409, None)
L['fast_none_bases'] = case(FastRGCNConv(N, 10, num_relations=5, num_bases=3), 410, None)
G['layers'] = L
out = os.path.join(HERE, 'golden_rgcn_v1.pt')
torch.save(G, out)
print('wrote', out, os.path.getsize(out), 'bytes')

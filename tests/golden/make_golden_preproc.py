"""Generates tests/golden/golden_preproc_v1.pt by running the REAL reference (PyG,
/root/reference) on CPU: sort_edge_index / coalesce / to_undirected / is_undirected
(SURVEY.md §8(f)-4).  Build container only:

    PYTHONPATH=/root/reference python tests/golden/make_golden_preproc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get('PYG_REFERENCE', '/root/reference'))
import torch_geometric  # noqa: E402
from torch_geometric.utils import (coalesce, is_undirected, sort_edge_index,  # noqa: E402
                                   to_undirected)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_preproc_v1.pt')
G = {'meta': {'torch': torch.__version__, 'pyg': torch_geometric.__version__}}


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ---- graphs ------------------------------------------------------------------------------------
n = 37
g = gen(50)
pairs = torch.randperm(n * n, generator=g)[:300]              # distinct (row, col) pairs
simple = torch.stack([pairs // n, pairs % n])
dup = torch.cat([simple[:, :200], simple[:, :120], simple[:, 50:90]], dim=1)
dup = dup[:, torch.randperm(dup.size(1), generator=g)]        # 360 edges, 200 distinct
attr_f = torch.randn(simple.size(1), 3, generator=g)
attr_i = torch.randint(0, 100, (simple.size(1), ), generator=g)
dattr = torch.randn(dup.size(1), 4, generator=g)
dattr[::5] = torch.randint(-2, 3, (dattr[::5].size(0), 4), generator=g).float()
dw = torch.rand(dup.size(1), generator=g) + 0.5

# ---- sort_edge_index ---------------------------------------------------------------------------
S = {'simple': simple, 'dup': dup, 'attr_f': attr_f, 'attr_i': attr_i, 'num_nodes': n}
for by_row in (True, False):
    ei, (af, ai) = sort_edge_index(simple, [attr_f, attr_i], n, sort_by_row=by_row)
    S[f'simple_by_row={by_row}'] = {'edge_index': ei, 'attr_f': af, 'attr_i': ai}
    S[f'dup_by_row={by_row}'] = sort_edge_index(dup, num_nodes=n, sort_by_row=by_row)
S['infer_num_nodes'] = sort_edge_index(simple)
G['sort_edge_index'] = S

# ---- coalesce ----------------------------------------------------------------------------------
C = {'dup': dup, 'attr': dattr, 'w': dw, 'num_nodes': n}
for red in ['sum', 'mean', 'min', 'max', 'mul', 'any']:
    ei, a = coalesce(dup, dattr, n, reduce=red)
    C[red] = {'edge_index': ei, 'attr': a}
ei, (a, w) = coalesce(dup, [dattr, dw], n, reduce='sum', sort_by_row=False)
C['list_by_col'] = {'edge_index': ei, 'attr': a, 'w': w}
C['no_attr'] = coalesce(dup, num_nodes=n)
ei, a = coalesce(dup, None, n)
C['none_attr'] = {'edge_index': ei, 'attr': a}
pre, pre_attr = sort_edge_index(dup, dattr, n)
ei, a = coalesce(pre, pre_attr, n, reduce='sum', is_sorted=True)
C['is_sorted'] = {'in_edge_index': pre, 'in_attr': pre_attr, 'edge_index': ei, 'attr': a}
ei, a = coalesce(simple, attr_f, n)                             # nothing to merge
C['simple'] = {'edge_index': ei, 'attr': a}
# gradient of the merged attributes
leaf = dattr.clone().requires_grad_(True)
_, a = coalesce(dup, leaf, n, reduce='mean')
go = torch.randn(a.shape, generator=gen(51))
C['mean_grad'] = {'grad_out': go, 'grad_attr': torch.autograd.grad(a, leaf, go)[0]}
G['coalesce'] = C

# ---- to_undirected / is_undirected -------------------------------------------------------------
U = {'edge_index': simple[:, :150], 'attr': attr_f[:150], 'num_nodes': n}
for red in ['add', 'mean', 'max']:
    ei, a = to_undirected(U['edge_index'], U['attr'], n, reduce=red)
    U[red] = {'edge_index': ei, 'attr': a}
U['no_attr'] = to_undirected(U['edge_index'])
und, und_attr = U['add']['edge_index'], U['add']['attr']
U['is_undirected'] = {
    'directed': is_undirected(U['edge_index'], num_nodes=n),
    'undirected': is_undirected(und, num_nodes=n),
    'undirected_attr': is_undirected(und, und_attr, n),
    'undirected_bad_attr': is_undirected(und, torch.arange(und.size(1)).float(), n),
}
G['undirected'] = U

torch.save(G, OUT)
print('wrote', OUT, os.path.getsize(OUT), 'bytes')
print(U['is_undirected'])

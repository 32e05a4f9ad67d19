"""Generates tests/golden/golden_khop_v1.pt by running the REAL reference (PyG, /root/reference)
on CPU: ``k_hop_subgraph(..., directed=True, flow='source_to_target')``
(utils/_subgraph.py:249-370) — the node set and the set of traversed edges that a neighbour
sampler with ``num_neighbors = [-1] * k`` must reproduce exactly.  Build container only:
    PYTHONPATH=/root/reference python tests/golden/make_golden_khop.py
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get('PYG_REFERENCE', '/root/reference'))
import torch_geometric  # noqa: E402
from torch_geometric.utils import k_hop_subgraph  # noqa: E402

g = torch.Generator().manual_seed(77)
N, E = 400, 2400
edge_index = torch.randint(0, N, (2, E), generator=g)
edge_index[:, :40] = edge_index[:, 40:80]          # duplicate edges
edge_index[1, 80:100] = edge_index[0, 80:100]      # self-loops
edge_index[1, 100:400] = 7                         # a hub destination
seeds = torch.randperm(N, generator=g)[:12]
out = {'meta': {'torch': torch.__version__, 'pyg': torch_geometric.__version__},
       'N': N, 'edge_index': edge_index, 'seeds': seeds, 'hops': {}}
for k in (1, 2, 3):
    subset, _, mapping, edge_mask = k_hop_subgraph(seeds, k, edge_index, num_nodes=N,
                                                   flow='source_to_target', directed=True)
    out['hops'][k] = {'subset': subset, 'edge_ids': edge_mask.nonzero().view(-1)}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_khop_v1.pt')
torch.save(out, path)
print(path, {k: (v['subset'].numel(), v['edge_ids'].numel()) for k, v in out['hops'].items()})

"""Generates tests/golden/golden_v1.pt by running the REAL reference (PyG, /root/reference) on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

Every entry holds the seeded inputs, the reference's outputs and (where differentiable) the
gradients for a fixed random ``grad_out``.  tests/test_oracle_golden.py pins ``oracle/pyg_oracle.py``
against these vectors; the ``-m gpu`` tests pin the HIP path against them.
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get('PYG_REFERENCE', '/root/reference'))
import torch_geometric  # noqa: E402
import torch_geometric.typing as pyg_typing  # noqa: E402
from torch_geometric.index import index2ptr, ptr2index  # noqa: E402
from torch_geometric.nn import (GAT, GCN, GATConv, GCNConv, GraphConv, GraphSAGE,  # noqa
                                RGCNConv, SAGEConv)
from torch_geometric.nn.conv.gcn_conv import gcn_norm  # noqa: E402
from torch_geometric.utils import (add_remaining_self_loops, add_self_loops, index_sort,  # noqa
                                   remove_self_loops, scatter, segment, softmax, spmm,
                                   to_torch_csc_tensor)
from torch_geometric.utils._scatter import scatter_argmax  # noqa: E402

assert not pyg_typing.WITH_TORCH_SCATTER and not pyg_typing.WITH_PYG_LIB, \
    "goldens must come from the plain CPU scatter path"

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden_v1.pt')
G = {'meta': {'torch': torch.__version__, 'pyg': torch_geometric.__version__}}


def gen(seed):
    return torch.Generator().manual_seed(seed)


def with_grad(fn, inputs, grad_out_seed):
    """Runs fn(*inputs) with requires_grad on float inputs; returns (out, grads)."""
    leaves = [t.clone().requires_grad_(True) if t.is_floating_point() else t for t in inputs]
    out = fn(*leaves)
    go = torch.randn(out.shape, generator=gen(grad_out_seed))
    fl = [t for t in leaves if t.is_floating_point()]
    grads = torch.autograd.grad(out, fl, go, allow_unused=True)
    return out.detach(), go, [None if g is None else g.detach() for g in grads]


# ---- scatter ---------------------------------------------------------------------------------------
g = gen(10)
src = torch.randn(60, 7, generator=g)
# ties and exact zeros so the min/max tie-splitting rule (and its zero-self quirk) is exercised
src[::3] = torch.randint(-2, 3, (20, 7), generator=g).float()
index = torch.randint(0, 12, (60, ), generator=g)
index[index == 5] = 4  # group 5 empty; dim_size 15 leaves 12..14 empty too
sc = {'src': src, 'index': index, 'dim_size': 15}
for red in ['sum', 'mean', 'min', 'max', 'mul']:
    out, go, (gs, ) = with_grad(lambda s: scatter(s, index, 0, 15, red), [src], 100)
    sc[red] = {'out': out, 'grad_out': go, 'grad_src': gs}
sc['any'] = scatter(src, index, 0, 15, 'any')
src3 = torch.randn(4, 30, 3, generator=g)
idx3 = torch.randint(0, 6, (30, ), generator=g)
out, go, (gs, ) = with_grad(lambda s: scatter(s, idx3, 1, 6, 'mean'), [src3], 101)
sc['dim1_mean'] = {'src': src3, 'index': idx3, 'out': out, 'grad_out': go, 'grad_src': gs}
out, go, (gs, ) = with_grad(lambda s: scatter(s, idx3, -2, None, 'max'), [src3], 102)
sc['dim1_max_nosize'] = {'out': out, 'grad_out': go, 'grad_src': gs}
v = torch.randn(25, generator=g)
i1 = torch.randint(0, 5, (25, ), generator=g)
sc['vec_sum'] = {'src': v, 'index': i1, 'out': scatter(v, i1, 0, None, 'sum')}
sc['argmax_known'] = {
    'src': torch.arange(5).float(), 'index': torch.tensor([2, 2, 0, 0, 3]), 'dim_size': 6,
    'out': scatter_argmax(torch.arange(5).float(), torch.tensor([2, 2, 0, 0, 3]), dim_size=6)}
va = torch.randint(0, 4, (40, ), generator=g).float()
ia = torch.randint(0, 9, (40, ), generator=g)
sc['argmax_rand'] = {'src': va, 'index': ia, 'dim_size': 11,
                     'out': scatter_argmax(va, ia, dim_size=11)}
G['scatter'] = sc

# ---- segment ---------------------------------------------------------------------------------------
g = gen(11)
ssrc = torch.randn(40, 5, generator=g)
ssrc[::4] = torch.randint(-1, 2, (10, 5), generator=g).float()
ptr = torch.tensor([0, 0, 5, 5, 17, 18, 40])  # empty first segment, empty middle segment
sg = {'src': ssrc, 'ptr': ptr}
for red in ['sum', 'mean', 'min', 'max']:
    out, go, (gs, ) = with_grad(lambda s: segment(s, ptr, red), [ssrc], 110)
    sg[red] = {'out': out, 'grad_out': go, 'grad_src': gs}
G['segment'] = sg

# ---- softmax ---------------------------------------------------------------------------------------
g = gen(12)
sm = {'known': {'src': torch.ones(4), 'index': torch.tensor([0, 0, 1, 2]),
                'ptr': torch.tensor([0, 2, 3, 4]),
                'out_index': softmax(torch.ones(4), torch.tensor([0, 0, 1, 2])),
                'out_ptr': softmax(torch.ones(4), None, torch.tensor([0, 2, 3, 4]))}}
a = torch.randn(50, 8, generator=g) * 3
sidx = torch.randint(0, 9, (50, ), generator=g).sort().values
sptr = index2ptr(sidx, 11)
out, go, (gs, ) = with_grad(lambda s: softmax(s, sidx, num_nodes=11), [a], 120)
sm['index'] = {'src': a, 'index': sidx, 'num_nodes': 11, 'out': out, 'grad_out': go,
               'grad_src': gs}
out2, go2, (gs2, ) = with_grad(lambda s: softmax(s, None, sptr), [a], 120)
sm['ptr'] = {'ptr': sptr, 'out': out2, 'grad_out': go2, 'grad_src': gs2}
perm = torch.randperm(50, generator=g)
out3, go3, (gs3, ) = with_grad(lambda s: softmax(s, sidx[perm], num_nodes=11), [a[perm]], 121)
sm['unsorted'] = {'src': a[perm], 'index': sidx[perm], 'out': out3, 'grad_out': go3,
                  'grad_src': gs3}
a3 = torch.randn(5, 50, generator=g)
out4, go4, (gs4, ) = with_grad(lambda s: softmax(s, sidx, num_nodes=11, dim=-1), [a3], 122)
sm['dim1'] = {'src': a3, 'out': out4, 'grad_out': go4, 'grad_src': gs4}
G['softmax'] = sm

# ---- integer goldens ---------------------------------------------------------------------------
g = gen(13)
keys = torch.randint(0, 37, (500, ), generator=g)
sk, sp = index_sort(keys, max_value=37, stable=True)
ts, tp = torch.sort(keys, stable=True)
assert torch.equal(sk, ts) and torch.equal(sp, tp)
from torch_geometric import EdgeIndex  # noqa: E402
ei_known = EdgeIndex([[0, 1, 1, 2], [1, 0, 2, 1]], sort_order='row').fill_cache_()
G['index'] = {
    'keys': keys, 'sorted': sk, 'perm': sp,
    'ptr': index2ptr(sk, 40), 'ptr_size': 40,
    'ptr2index': ptr2index(index2ptr(sk, 40)),
    'keys32_ptr': index2ptr(sk.int(), 40),
    'known_index2ptr': {'index': torch.tensor([0, 1, 1, 2]), 'ptr': index2ptr(
        torch.tensor([0, 1, 1, 2]), 3)},                       # test/test_index.py:85-96
    'known_edge_index': {'edge_index': torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]]),
                         'indptr': ei_known._indptr.clone(),   # test/test_edge_index.py:196-233
                         'T_indptr': ei_known._T_indptr.clone()},
}

# ---- graph used by the layer goldens -------------------------------------------------------------
g = gen(14)
N, E = 40, 300
edge_index = torch.randint(0, N - 1, (2, E), generator=g)  # node N-1 isolated
edge_index[:, :6] = torch.tensor([[0, 1, 2, 3, 3, 3], [0, 1, 5, 7, 7, 7]])  # loops + duplicates
edge_type = torch.randint(0, 5, (E, ), generator=g)
edge_weight = torch.rand(E, generator=g) + 0.1
x16 = torch.randn(N, 16, generator=g)
G['graph'] = {'N': N, 'edge_index': edge_index, 'edge_type': edge_type,
              'edge_weight': edge_weight, 'x': x16}

# spmm on a sparse adj_t (utils/_spmm.py) for the four reduces
# (to_torch_csc_tensor drops duplicate edges, so the case is stated on the coalesced list)
from torch_geometric.utils import coalesce  # noqa: E402
ei_c = coalesce(edge_index, num_nodes=N)
adj_t = to_torch_csc_tensor(ei_c, size=(N, N)).t()  # CSR with rows = destinations
sp_ = {'edge_index': ei_c}
for red in ['sum', 'mean', 'min', 'max']:
    if red in ('sum', 'mean'):
        out, go, (gx, ) = with_grad(lambda xx: spmm(adj_t, xx, red), [x16], 130)
        sp_[red] = {'out': out, 'grad_out': go, 'grad_x': gx}
    else:
        sp_[red] = {'out': spmm(adj_t, x16, red)}
G['spmm'] = sp_

# self-loop helpers + gcn_norm
ei_r, ew_r = add_remaining_self_loops(edge_index, edge_weight, 2.0, N)
ei_n, ew_n = gcn_norm(edge_index, edge_weight, N, False, True, 'source_to_target', torch.float32)
ei_n0, ew_n0 = gcn_norm(edge_index, None, N, False, True, 'source_to_target', torch.float32)
ei_rm, _ = remove_self_loops(edge_index)
ei_add, _ = add_self_loops(ei_rm, num_nodes=N)
G['loops'] = {'remaining_ei': ei_r, 'remaining_ew': ew_r, 'norm_ei': ei_n, 'norm_ew': ew_n,
              'norm0_ei': ei_n0, 'norm0_ew': ew_n0, 'gat_ei': ei_add}


# ---- layers ----------------------------------------------------------------------------------------
def layer_case(conv, seed, *args, **kwargs):
    conv.eval()
    xx = x16.clone().requires_grad_(True)
    out = conv(xx, *args, **kwargs)
    go = torch.randn(out.shape, generator=gen(seed))
    params = list(conv.parameters())
    grads = torch.autograd.grad(out, [xx] + params, go, allow_unused=True)
    return {'state': {k: v.detach().clone() for k, v in conv.state_dict().items()},
            'out': out.detach(), 'grad_out': go, 'grad_x': grads[0],
            'grad_params': {n: (None if gg is None else gg.detach())
                            for (n, _), gg in zip(conv.named_parameters(), grads[1:])}}


torch.manual_seed(20)
L = {}
L['sage_mean'] = layer_case(SAGEConv(16, 24, aggr='mean'), 200, edge_index)
L['sage_max'] = layer_case(SAGEConv(16, 24, aggr='max'), 201, edge_index)
L['sage_sum_noroot'] = layer_case(SAGEConv(16, 24, aggr='sum', root_weight=False, bias=False),
                                  202, edge_index)
L['gcn'] = layer_case(GCNConv(16, 12), 203, edge_index)
L['gcn_weighted'] = layer_case(GCNConv(16, 12), 204, edge_index, edge_weight)
L['gcn_nonorm'] = layer_case(GCNConv(16, 12, normalize=False), 205, edge_index, edge_weight)
L['gat'] = layer_case(GATConv(16, 6, heads=4), 206, edge_index)
L['gat_mean_heads'] = layer_case(GATConv(16, 6, heads=4, concat=False), 207, edge_index)
L['gat_noloops'] = layer_case(GATConv(16, 6, heads=2, add_self_loops=False), 208, edge_index)
L['rgcn'] = layer_case(RGCNConv(16, 10, num_relations=5), 209, edge_index, edge_type)
L['rgcn_blocks'] = layer_case(RGCNConv(16, 12, num_relations=5, num_blocks=4), 210, edge_index,
                              edge_type)
L['rgcn_bases'] = layer_case(RGCNConv(16, 10, num_relations=5, num_bases=3), 211, edge_index,
                             edge_type)
L['graph_add_weighted'] = layer_case(GraphConv(16, 10, aggr='add'), 212, edge_index, edge_weight)
L['graph_mean'] = layer_case(GraphConv(16, 10, aggr='mean'), 213, edge_index)
L['graph_max'] = layer_case(GraphConv(16, 10, aggr='max'), 214, edge_index)
G['layers'] = L

# GAT attention weights (return_attention_weights)
torch.manual_seed(21)
conv = GATConv(16, 6, heads=4).eval()
out, (ei_att, alpha) = conv(x16, edge_index, return_attention_weights=True)
G['gat_attention'] = {'state': {k: v.detach().clone() for k, v in conv.state_dict().items()},
                      'out': out.detach(), 'edge_index': ei_att, 'alpha': alpha.detach()}


# ---- models --------------------------------------------------------------------------------------
def model_case(model, seed, **kw):
    model.eval()
    xx = x16.clone().requires_grad_(True)
    out = model(xx, edge_index, **kw)
    go = torch.randn(out.shape, generator=gen(seed))
    params = list(model.parameters())
    grads = torch.autograd.grad(out, [xx] + params, go)
    return {'state': {k: v.detach().clone() for k, v in model.state_dict().items()},
            'out': out.detach(), 'grad_out': go, 'grad_x': grads[0],
            'grad_params': {n: gg.detach()
                            for (n, _), gg in zip(model.named_parameters(), grads[1:])}}


torch.manual_seed(22)
M = {}
M['graphsage'] = model_case(GraphSAGE(16, 32, num_layers=3, out_channels=8), 300)
M['gcn'] = model_case(GCN(16, 16, num_layers=2, out_channels=7), 301)
M['gat'] = model_case(GAT(16, 32, num_layers=3, out_channels=5, heads=4), 302)
G['models'] = M

torch.save(G, OUT)
print('wrote', OUT, os.path.getsize(OUT), 'bytes')

"""The REAL reference on the GPU box: ``oracle/_ref`` (staged by ``oracle/make_ref.py``, test
infrastructure) + ``pytorch_geometric_amd.backend.install()``.  The reference's OWN modules —
``nn.conv.{SAGEConv, GCNConv, GraphConv, GATConv}``, ``nn.models.GraphSAGE``, ``nn.aggr.*``,
``EdgeIndex.matmul``, ``utils.{scatter, segment, softmax, spmm, index_sort, sort_edge_index,
coalesce}`` — run on HIP tensors through the installed backend and are compared with the SAME
reference objects evaluated on the CPU (where the backend always steps aside).

This is the first execution of ``backend.py``'s device branches against the real reference
(round-1 VERDICT, weak #4): the CPU-side glue test (tests/test_backend_install.py) swaps the
kernels for the oracle."""
import copy

import pytest
import torch

from tests._util import assert_close, assert_close_scaled, gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pyg():
    from oracle import make_ref
    return make_ref.import_reference()  # ImportError = the staged reference did not travel


@pytest.fixture()
def installed(pyg):
    from pytorch_geometric_amd import backend, edge_index
    backend.install()
    edge_index.clear_cache()
    yield backend
    backend.uninstall()


@pytest.fixture()
def launches(monkeypatch):
    """Counts what reaches the HIP library: SpMM launches (with their widths) and radix sorts."""
    from pytorch_geometric_amd import _native
    rec = {'spmm': [], 'sorts': 0}
    sink = []
    monkeypatch.setattr(_native, 'timing_sink', sink)
    real_sort = _native.index_sort

    def counted_sort(*a, **k):
        rec['sorts'] += 1
        return real_sort(*a, **k)

    monkeypatch.setattr(_native, 'index_sort', counted_sort)
    rec['sink'] = sink
    return rec


def _fwd_bwd(module, args, grad_out, kwargs=None):
    for p in module.parameters():
        p.grad = None
    leaves = [a.clone().requires_grad_(True) if isinstance(a, torch.Tensor)
              and a.is_floating_point() else a for a in args]
    out = module(*leaves, **(kwargs or {}))
    out.backward(grad_out.to(out.device))
    grads = [a.grad for a in leaves if isinstance(a, torch.Tensor) and a.is_floating_point()]
    return out.detach(), grads, [p.grad.clone() for p in module.parameters()]


def _to(dev, args):
    return [a.to(dev) if isinstance(a, torch.Tensor) else a for a in args]


def test_reference_convs_take_the_hip_path_and_match_their_cpu_results(pyg, installed, launches,
                                                                       dev):
    from torch_geometric.nn import GATConv, GCNConv, GraphConv, SAGEConv
    g = gen(5)
    n, e = 300, 4000
    x = torch.randn(n, 16, generator=g)
    ei = torch.randint(0, n, (2, e), generator=g)
    # at most ONE self-loop per node: with duplicates the reference's add_remaining_self_loops
    # (`loop_attr[index] = edge_attr[...]`, utils/loop.py:640-644) keeps an unspecified one — its
    # own CPU and GPU results then differ (checked: 16 of 300 rows by up to 9e-2 on this graph)
    loops = (ei[0] == ei[1]).nonzero().view(-1)
    ei[1, loops] = (ei[1, loops] + 1) % n
    ei[:, :20] = torch.arange(20).repeat(2, 1) * 7
    w = torch.rand(e, generator=g)
    torch.manual_seed(0)
    cases = [
        ('sage-mean', SAGEConv(16, 12), (x, ei)),
        ('sage-max', SAGEConv(16, 12, aggr='max'), (x, ei)),
        ('sage-sum-t2s', SAGEConv(16, 12, aggr='sum', flow='target_to_source'), (x, ei)),
        ('gcn', GCNConv(16, 12), (x, ei)),
        ('gcn-weighted', GCNConv(16, 12, improved=True), (x, ei, w)),
        ('gcn-wide', GCNConv(16, 40), (x, ei)),   # input narrower: aggregated before lin
        ('graphconv', GraphConv(16, 12, aggr='mean'), (x, ei, w)),
        ('gat', GATConv(16, 4, heads=3), (x, ei)),
        ('gat-mean-heads', GATConv(16, 6, heads=2, concat=False), (x, ei)),
    ]
    for name, conv, args in cases:
        go = torch.randn(n, conv(*args).size(1), generator=g)
        ref_out, ref_gin, ref_gp = _fwd_bwd(conv, args, go)  # CPU: the reference's own path
        dconv = copy.deepcopy(conv).to(dev)
        before = len(launches['sink'])
        out, gin, gp = _fwd_bwd(dconv, _to(dev, args), go)
        assert len(launches['sink']) > before, f'{name}: no SpMM launch — reference path taken'
        if name == 'gcn-wide':  # forward and backward aggregation at the INPUT width
            fs = [i['F'] for i, _, _ in launches['sink'][before:] if 'F' in i]
            assert fs.count(16) >= 2 and 40 not in fs, fs
        assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what=f'{name} out')
        for a, b in zip(gin, ref_gin):
            assert_close(a, b, rtol=1e-5, atol=2e-5, what=f'{name} grad input')
        for a, b in zip(gp, ref_gp):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grad param')
    # the reference's SAGEConv.forward on a square graph = the one-kernel layer, one autograd node
    dconv = copy.deepcopy(cases[0][1]).to(dev)
    assert 'FusedSageStack' in dconv(x.to(dev), ei.to(dev)).grad_fn.name()
    gconv = GraphConv(16, 12).to(dev)   # (unweighted: the same layer under other names)
    assert 'FusedSageStack' in gconv(x.to(dev), ei.to(dev)).grad_fn.name()
    assert 'FusedSageStack' not in gconv(x.to(dev), ei.to(dev), w.to(dev)).grad_fn.name()
    assert_close(gconv(x.to(dev), ei.to(dev)), copy.deepcopy(gconv).cpu()(x, ei).detach(),
                 rtol=1e-5, atol=2e-5, what='graphconv unweighted')
    installed.uninstall()
    assert 'FusedSageStack' not in dconv(x.to(dev), ei.to(dev)).grad_fn.name()
    installed.install()


def test_reference_layers_do_not_resort_the_graph_every_forward(pyg, installed, launches, dev):
    """GCNConv(cached=False) and GATConv rebuild ``edge_index`` (self-loops) on every forward; the
    identity memos of install() hand the handle cache the same tensor again: the radix sorts
    happen once (by destination in the forward, by source in the first backward), not per step."""
    from torch_geometric.nn import GATConv, GCNConv
    g = gen(9)
    n = 500
    x = torch.randn(n, 8, generator=g).to(dev)
    ei = torch.randint(0, n, (2, 6000), generator=g).to(dev)
    torch.manual_seed(1)
    for conv in (GCNConv(8, 8).to(dev), GATConv(8, 4, heads=2).to(dev)):
        conv(x, ei).sum().backward()
        first = launches['sorts']
        assert first >= 1
        outs = []
        for _ in range(3):
            out = conv(x, ei)
            out.sum().backward()
            outs.append(out.detach())
        assert launches['sorts'] == first, (type(conv).__name__, launches['sorts'] - first)
        # (the reference's softmax / degree go through atomic scatter-adds: equal to the last ulp)
        assert_close(outs[0], outs[2], rtol=1e-6, atol=1e-6, what='repeatability')
    # an in-place edit of the input must invalidate the memo (tensor version counter)
    conv = GCNConv(8, 8).to(dev)
    a = conv(x, ei).detach()
    ei[0, :50] = (ei[0, :50] + 1) % n
    b = conv(x, ei).detach()
    ref = copy.deepcopy(conv).cpu()(x.cpu(), ei.cpu()).detach()
    assert not torch.equal(a, b)
    assert_close(b, ref, rtol=1e-5, atol=2e-5, what='gcn after in-place edit')


def test_reference_graphsage_model_runs_the_fused_stack(pyg, installed, launches, dev):
    """torch_geometric.nn.GraphSAGE + install() reaches the schedule of the headline number: one
    SpMM per layer and direction, the 64 -> 10 output layer aggregated at width 12 (narrow-layer
    re-order), not the per-layer propagate route."""
    from torch_geometric.nn import GraphSAGE
    g = gen(12)
    n = 400
    x = torch.randn(n, 20, generator=g)
    ei = torch.randint(0, n, (2, 5000), generator=g)
    torch.manual_seed(2)
    model = GraphSAGE(20, 64, num_layers=3, out_channels=10)
    go = torch.randn(n, 10, generator=g)
    ref_out, ref_gin, ref_gp = _fwd_bwd(model, (x, ei), go)
    dmodel = copy.deepcopy(model).to(dev)
    out, gin, gp = _fwd_bwd(dmodel, _to(dev, (x, ei)), go)
    # (the sink also carries the dense-transform launches, which have no 'F')
    widths = sorted(info['F'] for info, _, _ in launches['sink'] if 'F' in info)
    # six aggregation launches: forward at 20 / 64 / 12 (the 64 -> 10 layer re-ordered), backward
    # at 12, 64 and — the layer-1 input gradient — 20 as a transposed SpMM or 64 as the one-kernel
    # input gradient (which gathers at the layer's OUTPUT width)
    assert len(widths) == 6 and widths.count(12) == 2 and widths.count(20) >= 1 \
        and widths[-1] == 64, widths
    assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what='GraphSAGE out')
    assert_close_scaled(gin[0], ref_gin[0], tol=2e-5, what='GraphSAGE grad x')
    for a, b in zip(gp, ref_gp):
        assert_close_scaled(a, b, tol=2e-5, what='GraphSAGE grad param')
    # switched off: the reference's own layer loop runs, every SAGEConv.forward as a one-kernel
    # layer node (the same six aggregation widths, the ReLUs as ATen passes), same numbers
    dmodel.fuse_stack = False
    del launches['sink'][:]
    out2, _, _ = _fwd_bwd(dmodel, _to(dev, (x, ei)), go)
    assert sorted(i['F'] for i, _, _ in launches['sink'] if 'F' in i) == widths
    assert_close(out2, ref_out, rtol=1e-5, atol=2e-5, what='GraphSAGE layer loop')
    # ... and without the layer nodes: one fused propagate per layer at the layer's INPUT width
    from pytorch_geometric_amd.nn.models import _fused_sage
    _fused_sage.LAYER_NODE = False
    try:
        del launches['sink'][:]
        out3, _, _ = _fwd_bwd(dmodel, _to(dev, (x, ei)), go)
    finally:
        _fused_sage.LAYER_NODE = True
    assert sorted(i['F'] for i, _, _ in launches['sink'] if 'F' in i) == [20, 20, 64, 64, 64, 64]
    assert_close(out3, ref_out, rtol=1e-5, atol=2e-5, what='GraphSAGE propagate loop')


def test_reference_edge_index_matmul_on_device(pyg, installed, launches, dev):
    from torch_geometric import EdgeIndex
    g = gen(4)
    # square: the reference's own CPU backward of the transposed product builds its adjoint with
    # size=get_sparse_size()[::-1] (edge_index.py:1886-1894) and fails with a shape error on
    # non-square inputs — nothing to compare against there
    n_row = n_col = 110
    raw = torch.stack([torch.randint(0, n_row, (900, ), generator=g),
                       torch.randint(0, n_col, (900, ), generator=g)])
    value = torch.rand(900, generator=g)
    for order, transpose, other_rows in (('row', False, n_col), ('col', True, n_row)):
        srt = EdgeIndex(raw, sparse_size=(n_row, n_col)).sort_by(order)
        adj, val = srt.values, value[srt.indices]
        other = torch.randn(other_rows, 24, generator=g)
        for reduce in ('sum', 'mean', 'max', 'min'):
            for v in (None, val):
                if v is not None and reduce in ('max', 'min'):
                    continue
                o_cpu = other.clone().requires_grad_(True)
                ref = adj.matmul(o_cpu, v, reduce=reduce, transpose=transpose)
                ref.sum().backward()
                o_dev = other.to(dev).requires_grad_(True)
                before = len(launches['sink'])
                out = adj.to(dev).matmul(o_dev, None if v is None else v.to(dev), reduce=reduce,
                                         transpose=transpose)
                out.sum().backward()
                assert len(launches['sink']) > before, (order, reduce)
                assert_close(out, ref, rtol=1e-5, atol=2e-5, what=f'matmul {order} {reduce}')
                assert_close(o_dev.grad, o_cpu.grad, rtol=1e-5, atol=2e-5,
                             what=f'matmul grad {order} {reduce}')


def test_reference_utils_and_aggregations_on_device(pyg, installed, dev):
    import torch_geometric.utils as U
    from torch_geometric.nn import aggr
    g = gen(6)
    src = torch.randn(2000, 9, generator=g)
    index = torch.randint(0, 150, (2000, ), generator=g)
    for red in ('sum', 'mean', 'min', 'max', 'mul'):
        s = src if red != 'mul' else src.abs().add(0.5)
        ref = U.scatter(s, index, 0, 160, red)
        out = U.scatter(s.to(dev), index.to(dev), 0, 160, red)
        assert out.is_cuda
        assert_close(out, ref, rtol=2e-5, atol=2e-5, what=f'scatter {red}')
    sidx, _ = index.sort()
    ptr = torch._convert_indices_from_coo_to_csr(sidx, 160)
    for red in ('sum', 'mean', 'min', 'max'):
        assert_close(U.segment(src.to(dev), ptr.to(dev), red), U.segment(src, ptr, red),
                     rtol=1e-5, atol=2e-5, what=f'segment {red}')
    assert_close(U.softmax(src.to(dev), index.to(dev), num_nodes=160),
                 U.softmax(src, index, num_nodes=160), what='softmax index')
    assert_close(U.softmax(src.to(dev), ptr=ptr.to(dev)), U.softmax(src, ptr=ptr),
                 what='softmax ptr')
    keys = torch.randint(0, 1000, (5000, ), generator=g)
    v_ref, p_ref = U.index_sort(keys, stable=True)
    v, p = U.index_sort(keys.to(dev), max_value=3, stable=True)  # a WRONG hint must not matter
    assert torch.equal(v.cpu(), v_ref) and torch.equal(p.cpu(), p_ref)
    # distinct edges: the reference's CPU sort is NOT stable (index_sort(stable=False) ->
    # Tensor.sort), so the attribute order of duplicate edges is unspecified there
    pairs = torch.randperm(3600, generator=g)[:3000]
    ei = torch.stack([pairs // 60, pairs % 60])
    attr = torch.randn(3000, 3, generator=g)
    r_ei, r_at = U.sort_edge_index(ei, attr, num_nodes=60)
    d_ei, d_at = U.sort_edge_index(ei.to(dev), attr.to(dev), num_nodes=60)
    assert torch.equal(d_ei.cpu(), r_ei)
    assert_close(d_at, r_at, rtol=0, atol=0, what='sort_edge_index attr')
    dup = torch.cat([ei, ei[:, :700]], dim=1)  # coalesce merges duplicates: order-free result
    dat = torch.cat([attr, attr[:700] + 1.0])
    r_ei, r_at = U.coalesce(dup, dat, num_nodes=60, reduce='mean')
    d_ei, d_at = U.coalesce(dup.to(dev), dat.to(dev), num_nodes=60, reduce='mean')
    assert torch.equal(d_ei.cpu(), r_ei)
    assert_close(d_at, r_at, rtol=1e-5, atol=1e-5, what='coalesce attr')
    adj = torch.sparse_coo_tensor(ei, torch.rand(3000, generator=g), (60, 60)).coalesce()
    dense = torch.randn(60, 7, generator=g)
    for red in ('sum', 'mean'):
        assert_close(U.spmm(adj.to_sparse_csr().to(dev), dense.to(dev), red),
                     U.spmm(adj.to_sparse_csr(), dense, red), rtol=1e-5, atol=2e-5,
                     what=f'spmm {red}')
    for mod in (aggr.MeanAggregation(), aggr.MaxAggregation(), aggr.SumAggregation(),
                aggr.StdAggregation(), aggr.MultiAggregation(['mean', 'max', 'std']),
                aggr.SoftmaxAggregation(t=0.5)):
        ref = mod(src, index, dim_size=160)
        out = mod(src.to(dev), index.to(dev), dim_size=160)
        assert_close(out, ref, rtol=2e-5, atol=2e-5, what=type(mod).__name__)
    # the reference's own wrapper converts an error raised AT the call (nn/aggr/base.py:131-141):
    # that needs the blocking flag read
    from pytorch_geometric_amd import _native
    _native.INDEX_CHECK = 'sync'
    try:
        with pytest.raises(ValueError, match="invalid 'dim_size'"):
            aggr.MeanAggregation()(src.to(dev), index.to(dev), dim_size=3)
    finally:
        _native.INDEX_CHECK = 'async'
    # the reference edits scatter outputs in place (gcn_conv.py:109 `deg.pow_(-0.5)`): outputs of
    # the custom autograd Functions must not be views
    w = torch.rand(2000, generator=g).to(dev).requires_grad_(True)
    deg = U.scatter(w, index.to(dev), 0, 160, 'sum')
    deg.pow_(-0.5)
    deg.sum().backward()
    w_c = w.detach().cpu().requires_grad_(True)
    ref = U.scatter(w_c, index, 0, 160, 'sum').pow(-0.5)
    ref.sum().backward()
    assert_close(w.grad, w_c.grad, rtol=2e-5, atol=2e-5, what='in-place edit of a scatter result')


def test_reference_node_loader_drives_the_gpu_sampler(pyg, installed, dev):
    """backend.neighbor_sampler(...) IS a torch_geometric.sampler.BaseSampler: the reference's
    NodeLoader (loader/node_loader.py:90-207) calls sample_from_nodes(NodeSamplerInput) and joins
    features with its own filter_data; the batch has the reference's fields."""
    from torch_geometric.data import Data
    from torch_geometric.loader import NodeLoader
    from torch_geometric.sampler import BaseSampler, SamplerOutput
    g = gen(21)
    N = 1500
    ei = torch.randint(0, N, (2, 20000), generator=g)
    x = torch.randn(N, 12, generator=g)
    y = torch.randint(0, 5, (N, ), generator=g)
    data = Data(x=x, y=y, edge_index=ei, num_nodes=N).to(dev)
    sampler = installed.neighbor_sampler(data, [6, 3], seed=5)
    assert isinstance(sampler, BaseSampler)
    train_idx = torch.randperm(N, generator=g)[:200]
    loader = NodeLoader(data, node_sampler=sampler, input_nodes=train_idx, batch_size=64,
                        shuffle=False)
    seen = 0
    for i, batch in enumerate(loader):
        seeds = train_idx[i * 64:(i + 1) * 64]
        bs = seeds.numel()
        assert batch.batch_size == bs and torch.equal(batch.input_id.cpu(),
                                                      torch.arange(i * 64, i * 64 + bs))
        n_id, e_id = batch.n_id.cpu(), batch.e_id.cpu()
        assert torch.equal(n_id[:bs], seeds)
        assert torch.equal(batch.x.cpu(), x[n_id]) and torch.equal(batch.y.cpu(), y[n_id])
        loc = batch.edge_index.cpu()
        assert torch.equal(n_id[loc[0]], ei[0][e_id]) and torch.equal(n_id[loc[1]], ei[1][e_id])
        assert len(batch.num_sampled_nodes) == 3 and len(batch.num_sampled_edges) == 2
        assert sum(batch.num_sampled_nodes) == n_id.numel()
        assert sum(batch.num_sampled_edges) == e_id.numel()
        # per-destination counts: min(in-degree, fan-out) for the seeds
        deg = torch.bincount(ei[1], minlength=N)
        first_hop = loc[1][:batch.num_sampled_edges[0]]
        got = torch.bincount(first_hop, minlength=bs)[:bs]
        assert torch.equal(got, deg[seeds].clamp(max=6))
        seen += bs
    assert seen == 200
    out = sampler.sample_from_nodes(loader.input_data[[0, 1, 2]])
    assert isinstance(out, SamplerOutput) and out.metadata[0].tolist() == [0, 1, 2]
    # the reference's model consumes the batch (trim_to_layer path included)
    from torch_geometric.nn import GraphSAGE
    torch.manual_seed(0)
    model = GraphSAGE(12, 16, num_layers=2, out_channels=5).to(dev)
    out = model(batch.x, batch.edge_index, num_sampled_nodes_per_hop=batch.num_sampled_nodes,
                num_sampled_edges_per_hop=batch.num_sampled_edges)
    assert out.shape[1] == 5 and bool(torch.isfinite(out).all())


def test_reference_node_loader_with_sampler_options(pyg, installed, dev):
    """The options NeighborLoader forwards to its sampler (loader/neighbor_loader.py:209-233), on
    the BaseSampler adapter, driven by the reference's NodeLoader: `disjoint=True` fills
    `batch.batch` (the reference's filter_data copies SamplerOutput.batch), `replace=True` gives
    exactly k edges per seed with an in-neighbour, `subgraph_type='bidirectional'` (string or the
    reference's SubgraphType) returns symmetric edges without hop counts."""
    from torch_geometric.data import Data
    from torch_geometric.loader import NodeLoader
    from torch_geometric.sampler.base import SubgraphType
    g = gen(33)
    N = 1200
    ei = torch.randint(0, N, (2, 15000), generator=g)
    x = torch.randn(N, 6, generator=g)
    data = Data(x=x, edge_index=ei, num_nodes=N).to(dev)
    seeds = torch.randperm(N, generator=g)[:96]
    deg = torch.bincount(ei[1], minlength=N)

    def batches(**opts):
        sampler = installed.neighbor_sampler(data, [4, 2], seed=3, **opts)
        return list(NodeLoader(data, node_sampler=sampler, input_nodes=seeds, batch_size=32,
                               shuffle=False))

    for b in batches(disjoint=True):
        tree = b.batch.cpu()
        n_id, loc = b.n_id.cpu(), b.edge_index.cpu()
        assert torch.equal(tree[:32], torch.arange(32)) and tree.numel() == n_id.numel()
        assert torch.equal(tree[loc[0]], tree[loc[1]])             # edges stay inside their tree
        assert (tree * N + n_id).unique().numel() == n_id.numel()
        assert torch.equal(b.x.cpu(), x[n_id])
    for i, b in enumerate(batches(replace=True)):
        s = seeds[i * 32:(i + 1) * 32]
        first_hop = b.edge_index.cpu()[1][:b.num_sampled_edges[0]]
        got = torch.bincount(first_hop, minlength=32)[:32]
        assert torch.equal(got, torch.where(deg[s] > 0, torch.full_like(deg[s], 4), deg[s]))
    for kind in ('bidirectional', SubgraphType.bidirectional):
        for b in batches(subgraph_type=kind):
            loc = b.edge_index.cpu()
            m = b.n_id.numel()
            fwd_keys = set((loc[1] * m + loc[0]).tolist())
            assert fwd_keys == set((loc[0] * m + loc[1]).tolist())  # every edge has its reverse
            assert 'num_sampled_edges' not in b or b.num_sampled_edges is None
    for b in batches(subgraph_type='induced'):   # every graph edge between the batch's nodes
        nid = b.n_id.cpu()
        inb = torch.zeros(N, dtype=torch.bool)
        inb[nid] = True
        want = int((inb[ei[0]] & inb[ei[1]]).sum())
        assert b.edge_index.size(1) == want and b.e_id.numel() == want
        loc = b.edge_index.cpu()
        assert torch.equal(nid[loc[0]], ei[0, b.e_id.cpu()])
        assert torch.equal(nid[loc[1]], ei[1, b.e_id.cpu()])


@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
@pytest.mark.parametrize('transpose', [False, True])
def test_reference_torch_sparse_route_lands_on_this_backend(pyg, dev, monkeypatch, reduce,
                                                            transpose):
    """The reference's `_torch_sparse_spmm` (edge_index.py:1768-1810) — the route it takes on the
    GPU when `torch_geometric.typing.WITH_TORCH_SPARSE` is on — calls
    `torch.ops.torch_sparse.spmm_{sum,mean,min,max}`; with torch-sparse absent those names are
    served by pytorch_geometric_amd/torch_sparse_ops.py.  NO backend.install() here: only the
    operator names connect the two.  Values and gradients against the reference's CPU route."""
    from pytorch_geometric_amd import torch_sparse_ops
    if torch_sparse_ops.torch_sparse_present():
        pytest.skip('the real torch-sparse owns the namespace in this environment')
    assert torch_sparse_ops.register() is True
    import torch_geometric.typing as pyg_typing
    from torch_geometric import EdgeIndex
    g = gen(12 + len(reduce))
    n_src, n_dst, e = 300, 260, 4000
    row = torch.randint(0, n_dst, (e, ), generator=g)
    col = torch.randint(0, n_src, (e, ), generator=g)
    if transpose:   # (n_src x n_dst)^T @ [n_src, F]: sorted by column
        ei = torch.stack([col, row])
        ei = ei[:, torch.argsort(ei[1] * n_src + ei[0])]
        size, order = (n_src, n_dst), 'col'
    else:
        ei = torch.stack([row, col])
        ei = ei[:, torch.argsort(ei[0] * n_src + ei[1])]
        size, order = (n_dst, n_src), 'row'
    x = torch.randn(n_src, 24, generator=g)
    val = torch.rand(e, generator=g) if reduce in ('sum', 'mean') else None
    go = torch.randn(n_dst, 24, generator=g)

    def run(device, with_ts):
        monkeypatch.setattr(pyg_typing, 'WITH_TORCH_SPARSE', with_ts)
        adj = EdgeIndex(ei.to(device), sparse_size=size, sort_order=order)
        xr = x.to(device).requires_grad_(True)
        v = None if val is None else val.to(device).requires_grad_(True)
        out = adj.matmul(xr, v, reduce=reduce, transpose=transpose)
        out.backward(go.to(device))
        return out.detach().cpu(), xr.grad.cpu(), None if v is None else v.grad.cpu()

    sink = []
    from pytorch_geometric_amd import _native
    monkeypatch.setattr(_native, 'timing_sink', sink)
    got = run(dev, True)
    monkeypatch.setattr(_native, 'timing_sink', None)
    assert any(info.get('reduce') == reduce for info, *_ in sink), 'no HIP SpMM launch was seen'
    want = run('cpu', False)
    assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what=f'{reduce} out')
    assert_close(got[1], want[1], rtol=1e-5, atol=2e-5, what=f'{reduce} grad mat')
    if val is not None:
        assert_close(got[2], want[2], rtol=1e-5, atol=5e-5, what=f'{reduce} grad value')


def test_reference_convs_accept_this_packages_edge_index_handle(pyg, installed, launches, dev):
    """The handle is a Tensor subclass (as the reference's EdgeIndex is): the reference's layers
    take it where they take ``edge_index``; SAGEConv / GraphConv aggregate on its cached sorted
    forms (no sort per call), GCNConv / GATConv see a plain tensor after their own rewrite."""
    import pytorch_geometric_amd as pga
    from torch_geometric.nn import GATConv, GCNConv, GraphConv, SAGEConv
    g = gen(11)
    n, e = 257, 3000
    x = torch.randn(n, 16, generator=g).to(dev)
    ei = torch.randint(0, n, (2, e), generator=g).to(dev)
    handle = pga.EdgeIndex(ei.clone(), (n, n))
    assert isinstance(handle, torch.Tensor) and type(handle[0]) is torch.Tensor
    torch.manual_seed(0)
    for name, conv in [('sage', SAGEConv(16, 12)), ('graphconv', GraphConv(16, 12, aggr='max')),
                       ('gcn', GCNConv(16, 12)), ('gat', GATConv(16, 4, heads=2))]:
        conv = conv.to(dev)
        go = torch.randn(n, conv(x, ei).size(1), generator=g)
        want = _fwd_bwd(conv, (x, ei), go)
        before, sorts = len(launches['sink']), launches['sorts']
        got = _fwd_bwd(conv, (x, handle), go)
        assert len(launches['sink']) > before, f'{name}: no SpMM launch with the handle'
        if name in ('sage', 'graphconv'):
            _fwd_bwd(conv, (x, handle), go)
            assert launches['sorts'] - sorts <= 2, f'{name}: the handle was re-sorted per call'
        assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what=f'{name} out')
        for a, b in zip(got[1] + got[2], want[1] + want[2]):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grads')


# ---- round 6: the rest of the drop-in boundary (VERDICT r5 "what's missing" 1-4) ----------------
@pytest.fixture()
def segmm_calls(monkeypatch):
    """Counts the grouped-GEMM launches that reach the HIP library."""
    from pytorch_geometric_amd import _native
    rec = {'fwd': 0, 'wgrad': 0}
    real, real_w = _native.segment_matmul, _native.segment_matmul_wgrad

    def fwd(*a, **k):
        rec['fwd'] += 1
        return real(*a, **k)

    def wgrad(*a, **k):
        rec['wgrad'] += 1
        return real_w(*a, **k)

    monkeypatch.setattr(_native, 'segment_matmul', fwd)
    monkeypatch.setattr(_native, 'segment_matmul_wgrad', wgrad)
    return rec


def _relational_graph(n, e, n_rel, seed):
    g = gen(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    # a skewed relation histogram with empty relations (FB15k-237 has both)
    et = (torch.rand(e, generator=g).pow(3) * (n_rel - 2)).long()
    return g, ei, et


def test_reference_rgcn_layers_run_the_segmented_schedule(pyg, installed, segmm_calls, dev):
    """torch_geometric.nn.{RGCNConv, FastRGCNConv} + install(): the reference's own classes leave
    their per-relation Python loop (rgcn_conv.py:243-282) for the sorted, segmented schedule — one
    grouped GEMM forward, one per gradient — and return what the loop returns on the CPU: dense,
    basis-decomposed and block-diagonal weights, mean / add / max, feature and node-index inputs,
    (source, destination) feature pairs."""
    from torch_geometric.nn import FastRGCNConv, RGCNConv
    n, e, n_rel = 220, 5000, 13
    g, ei, et = _relational_graph(n, e, n_rel, 41)
    x = torch.randn(n, 24, generator=g)
    torch.manual_seed(3)
    cases = [
        ('dense-mean', RGCNConv(24, 16, n_rel), (x, ei, et)),
        ('dense-add', RGCNConv(24, 16, n_rel, aggr='add'), (x, ei, et)),
        ('dense-max', RGCNConv(24, 16, n_rel, aggr='max'), (x, ei, et)),
        ('bases', RGCNConv(24, 16, n_rel, num_bases=4), (x, ei, et)),
        ('blocks', RGCNConv(24, 16, n_rel, num_blocks=4), (x, ei, et)),
        ('no-root-no-bias', RGCNConv(24, 16, n_rel, root_weight=False, bias=False), (x, ei, et)),
        ('fast-dense', FastRGCNConv(24, 16, n_rel), (x, ei, et)),
        ('fast-blocks-add', FastRGCNConv(24, 16, n_rel, num_blocks=2, aggr='add'), (x, ei, et)),
        ('fast-bases', FastRGCNConv(24, 16, n_rel, num_bases=3), (x, ei, et)),
    ]
    for name, conv, args in cases:
        go = torch.randn(n, 16, generator=g)
        ref_out, ref_gin, ref_gp = _fwd_bwd(conv, args, go)   # CPU: the reference's own loop
        dconv = copy.deepcopy(conv).to(dev)
        before = dict(segmm_calls)
        out, gin, gp = _fwd_bwd(dconv, _to(dev, args), go)
        assert segmm_calls['fwd'] - before['fwd'] == 2, (name, segmm_calls)   # forward + d input
        assert segmm_calls['wgrad'] - before['wgrad'] == 1, (name, segmm_calls)
        assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what=f'{name} out')
        assert_close(gin[0], ref_gin[0], rtol=1e-5, atol=2e-5, what=f'{name} grad x')
        for a, b in zip(gp, ref_gp):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grad param')
    # node-index ("featureless") inputs: x = None -> one embedding row per (relation, node)
    for name, conv in [('index', RGCNConv(n, 16, n_rel)), ('index-bases', RGCNConv(n, 16, n_rel, num_bases=3)),
                       ('fast-index', FastRGCNConv(n, 16, n_rel, aggr='add'))]:
        go = torch.randn(n, 16, generator=g)
        ref_out, _, ref_gp = _fwd_bwd(conv, (None, ei, et), go)
        dconv = copy.deepcopy(conv).to(dev)
        out, _, gp = _fwd_bwd(dconv, (None, ei.to(dev), et.to(dev)), go)
        assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what=f'{name} out')
        for a, b in zip(gp, ref_gp):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grad param')
    # (source, destination) pairs: a bipartite relation graph
    n_dst = 90
    ei2 = torch.stack([ei[0], ei[1] % n_dst])
    xs, xd = torch.randn(n, 24, generator=g), torch.randn(n_dst, 10, generator=g)
    conv = RGCNConv((24, 10), 16, n_rel)
    go = torch.randn(n_dst, 16, generator=g)
    ref_out, ref_gin, ref_gp = _fwd_bwd(conv, ((xs, xd), ei2, et), go)
    dconv = copy.deepcopy(conv).to(dev)
    before = segmm_calls['fwd']
    out, _, gp = _fwd_bwd(dconv, ((xs.to(dev), xd.to(dev)), ei2.to(dev), et.to(dev)), go)
    assert segmm_calls['fwd'] > before
    assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what='bipartite out')
    for a, b in zip(gp, ref_gp):
        assert_close_scaled(a, b, tol=2e-5, what='bipartite grad param')
    # a subclass that overrides `message` keeps the reference's loop (no grouped GEMM)

    class Scaled(RGCNConv):
        def message(self, x_j, edge_type_ptr):
            return 2.0 * x_j

    torch.manual_seed(4)
    conv = Scaled(24, 16, 4)
    et4 = et % 4
    ref = conv(x, ei, et4).detach()
    before = segmm_calls['fwd']
    out = copy.deepcopy(conv).to(dev)(x.to(dev), ei.to(dev), et4.to(dev))
    assert segmm_calls['fwd'] == before
    assert_close(out, ref, rtol=1e-5, atol=2e-5, what='overridden message')


def test_reference_hetero_linear_is_one_grouped_gemm(pyg, installed, segmm_calls, dev):
    """torch_geometric.nn.HeteroLinear + install() (nn/dense/linear.py:287-329): the per-type
    Python loop becomes sort -> one grouped GEMM -> restore, for sorted and unsorted type vectors
    and with empty types; pinned against the reference's `forward_naive` route on the CPU."""
    from torch_geometric.nn import HeteroLinear
    g = gen(17)
    n, n_types = 3000, 7
    x = torch.randn(n, 20, generator=g)
    tv = torch.randint(0, n_types - 1, (n, ), generator=g)   # the last type stays empty
    go = torch.randn(n, 12, generator=g)
    torch.manual_seed(5)
    for name, mod, types in [('unsorted', HeteroLinear(20, 12, n_types), tv),
                             ('sorted', HeteroLinear(20, 12, n_types, is_sorted=True),
                              tv.sort().values),
                             ('no-bias', HeteroLinear(20, 12, n_types, bias=False), tv)]:
        ref_out, ref_gin, ref_gp = _fwd_bwd(mod, (x, types), go)
        dmod = copy.deepcopy(mod).to(dev)
        before = segmm_calls['fwd']
        out, gin, gp = _fwd_bwd(dmod, (x.to(dev), types.to(dev)), go)
        assert segmm_calls['fwd'] - before == 2, name
        assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what=f'{name} out')
        assert_close(gin[0], ref_gin[0], rtol=1e-5, atol=2e-5, what=f'{name} grad x')
        for a, b in zip(gp, ref_gp):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grad param')
    # seam S2 by name: the reference's two call sites resolve `pyg_lib.ops.segment_matmul` here
    import torch_geometric.nn.conv.rgcn_conv as rgcn_mod
    import torch_geometric.nn.dense.linear as lin_mod
    ptr = torch.tensor([0, 5, 5, 40, 100])
    w = torch.randn(4, 20, 6, generator=g)
    want = torch.cat([x[ptr[i]:ptr[i + 1]] @ w[i] for i in range(4)])
    for mod in (rgcn_mod, lin_mod):
        got = mod.pyg_lib.ops.segment_matmul(x[:100].to(dev), ptr.to(dev), w.to(dev))
        assert_close(got, want, rtol=1e-5, atol=2e-5, what='pyg_lib.ops.segment_matmul')
    with pytest.raises(NotImplementedError):   # no host computation behind the name
        rgcn_mod.pyg_lib.ops.segment_matmul(x[:100], ptr, w)


def test_custom_message_passing_layer_gathers_through_the_hip_kernel(pyg, installed, dev):
    """A user's MessagePassing subclass with `message(x_i, x_j, edge_attr)`: install() serves the
    gather behind every `_i` / `_j` argument (`MessagePassing._index_select`,
    message_passing.py:263-290) and the scatter behind the aggregation; out-of-range indices
    raise the reference's IndexError texts."""
    from torch_geometric.nn import MessagePassing

    class EdgeConvLike(MessagePassing):
        def __init__(self, aggr):
            super().__init__(aggr=aggr)
            self.lin = torch.nn.Linear(2 * 8 + 3, 8)

        def forward(self, x, edge_index, edge_attr):
            return self.propagate(edge_index, x=x, edge_attr=edge_attr)

        def message(self, x_i, x_j, edge_attr):
            return self.lin(torch.cat([x_i, x_j - x_i, edge_attr], dim=-1))

    g = gen(8)
    n, e = 400, 6000
    x = torch.randn(n, 8, generator=g)
    ei = torch.randint(0, n, (2, e), generator=g)
    ea = torch.randn(e, 3, generator=g)
    for aggr in ('add', 'max', 'mean'):
        torch.manual_seed(1)
        conv = EdgeConvLike(aggr)
        go = torch.randn(n, 8, generator=g)
        ref_out, ref_gin, ref_gp = _fwd_bwd(conv, (x, ei, ea), go)
        dconv = copy.deepcopy(conv).to(dev)
        out, gin, gp = _fwd_bwd(dconv, _to(dev, (x, ei, ea)), go)
        assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what=f'{aggr} out')
        for a, b in zip(gin, ref_gin):
            assert_close(a, b, rtol=1e-5, atol=5e-5, what=f'{aggr} grad input')
        for a, b in zip(gp, ref_gp):
            assert_close_scaled(a, b, tol=2e-5, what=f'{aggr} grad param')

    # the gather itself is this package's autograd node (and only for float32 device rows)
    seen = []

    class Probe(MessagePassing):
        def forward(self, x, edge_index):
            return self.propagate(edge_index, x=x)

        def message(self, x_j):
            seen.append(x_j.grad_fn.name() if x_j.grad_fn is not None else None)
            return x_j

    probe = Probe(aggr='add')
    xd = x.to(dev).requires_grad_(True)
    probe(xd, ei.to(dev))
    probe(x.clone().requires_grad_(True), ei)
    assert 'GatherFunction' in seen[0] and 'GatherFunction' not in seen[1], seen
    installed.uninstall()
    probe(xd, ei.to(dev))
    installed.install()
    assert 'GatherFunction' not in seen[2], seen

    from pytorch_geometric_amd import _native
    bad = ei.clone()
    bad[0, 7] = n + 3
    neg = ei.clone()
    neg[1, 9] = -1
    _native.INDEX_CHECK = 'sync'
    try:
        with pytest.raises(IndexError, match=f'larger than {n - 1}'):
            probe(xd, bad.to(dev))
        with pytest.raises(IndexError, match='negative indices'):
            EdgeConvLike('add').to(dev)(xd, neg.to(dev), ea.to(dev))
    finally:
        _native.INDEX_CHECK = 'async'
    # default mode: the flag arrives behind the launch (as the reference's device assert does)
    # — at the next checked launch (the scatter of the same propagate, when the flag has already
    # landed) or at check_index_errors() at the latest
    with pytest.raises(IndexError, match=f'larger than {n - 1}'):
        probe(xd, bad.to(dev))
        _native.check_index_errors()


def test_reference_edge_index_as_the_graph_argument(pyg, installed, launches, dev):
    """The reference's own EdgeIndex (edge_index.py:173) handed to its own layers: SAGE / GCN /
    GAT / GraphConv reach the fused kernels on the order and caches it carries — sorted by
    destination: no sort for the forward; `fill_cache_()` done: no sort at all."""
    from torch_geometric import EdgeIndex
    from torch_geometric.nn import GATConv, GCNConv, GraphConv, GraphSAGE, SAGEConv
    g = gen(23)
    n, e = 350, 5000
    x = torch.randn(n, 16, generator=g)
    raw = torch.randint(0, n, (2, e), generator=g)
    loops = (raw[0] == raw[1]).nonzero().view(-1)
    raw[1, loops] = (raw[1, loops] + 1) % n
    by_col = EdgeIndex(raw, sparse_size=(n, n)).sort_by('col').values
    by_row = EdgeIndex(raw, sparse_size=(n, n)).sort_by('row').values
    unsorted = EdgeIndex(raw, sparse_size=(n, n))
    torch.manual_seed(6)
    layers = [('sage', SAGEConv(16, 12)), ('sage-max', SAGEConv(16, 12, aggr='max')),
              ('graphconv', GraphConv(16, 12)), ('gcn', GCNConv(16, 12)),
              ('gcn-raw', GCNConv(16, 12, normalize=False)), ('gat', GATConv(16, 4, heads=2))]
    for name, conv in layers:
        go = torch.randn(n, conv(x, raw).size(1), generator=g)
        for kind, adj in (('col', by_col), ('row', by_row), ('none', unsorted)):
            want = _fwd_bwd(conv, (x, adj.as_tensor()), go)       # CPU, plain tensor
            dconv = copy.deepcopy(conv).to(dev)
            dadj = adj.to(dev)
            assert type(dadj).__name__ == 'EdgeIndex' and dadj.is_cuda
            before = len(launches['sink'])
            got = _fwd_bwd(dconv, (x.to(dev), dadj), go)
            assert len(launches['sink']) > before, f'{name}/{kind}: no SpMM launch'
            assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what=f'{name}/{kind} out')
            for a, b in zip(got[1] + got[2], want[1] + want[2]):
                assert_close_scaled(a, b, tol=2e-5, what=f'{name}/{kind} grads')
    # the one-kernel layer is reached with the reference's graph object too
    dconv = copy.deepcopy(layers[0][1]).to(dev)
    assert 'FusedSageStack' in dconv(x.to(dev), by_col.to(dev)).grad_fn.name()
    # sort counts: destination-sorted input = no sort forward, one (by source) for the backward
    from pytorch_geometric_amd import edge_index as own_ei
    own_ei.clear_cache()
    dadj, xd = by_col.to(dev), x.to(dev).requires_grad_(True)
    conv = SAGEConv(16, 12, aggr='max').to(dev)    # (max: the propagate route, not the layer node)
    s0 = launches['sorts']
    out = conv(xd, dadj)
    assert launches['sorts'] == s0, 'a destination-sorted EdgeIndex was sorted again'
    out.sum().backward()
    s1 = launches['sorts']
    assert s1 - s0 <= 1
    conv(xd, dadj).sum().backward()
    assert launches['sorts'] == s1, 'the adopted handle was not found again'
    # ... and with the reference's cache filled beforehand, none at all
    own_ei.clear_cache()
    dadj = by_col.to(dev)
    dadj.fill_cache_()
    s2 = launches['sorts']
    conv(xd, dadj).sum().backward()
    assert launches['sorts'] == s2, 'the cached transposed form of the EdgeIndex was not used'
    ref = copy.deepcopy(conv).cpu()(x, by_col.as_tensor()).detach()
    assert_close(conv(xd, dadj), ref, rtol=1e-5, atol=2e-5, what='seeded handle')
    # the whole model
    torch.manual_seed(7)
    model = GraphSAGE(16, 32, num_layers=2, out_channels=5)
    go = torch.randn(n, 5, generator=g)
    want = _fwd_bwd(model, (x, raw), go)
    got = _fwd_bwd(copy.deepcopy(model).to(dev), (x.to(dev), by_col.to(dev)), go)
    assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what='GraphSAGE on EdgeIndex')
    for a, b in zip(got[2], want[2]):
        assert_close_scaled(a, b, tol=2e-5, what='GraphSAGE on EdgeIndex grads')


def test_reference_gatconv_runs_the_fused_attention_node(pyg, installed, launches, dev):
    """torch_geometric.nn.GATConv + install(): projection + ONE autograd node (node terms, edge
    softmax, aggregation) when nothing in between is observable; dropout on the coefficients,
    `return_attention_weights`, edge features or bipartite inputs keep the reference's own
    sequence (whose pieces are served too)."""
    from torch_geometric.nn import GAT, GATConv
    g = gen(31)
    n, e = 500, 7000
    x = torch.randn(n, 16, generator=g)
    ei = torch.randint(0, n, (2, e), generator=g)
    loops = (ei[0] == ei[1]).nonzero().view(-1)
    ei[1, loops] = (ei[1, loops] + 1) % n
    torch.manual_seed(8)

    def nodes(out):
        seen, todo = set(), [out.grad_fn]
        while todo:
            f = todo.pop()
            if f is None or f in seen:
                continue
            seen.add(f)
            todo += [nf for nf, _ in f.next_functions]
        return {f.name() for f in seen}

    for name, conv in [('concat', GATConv(16, 8, heads=4)),
                       ('mean-heads', GATConv(16, 8, heads=3, concat=False)),
                       ('residual', GATConv(16, 8, heads=2, residual=True)),
                       ('no-loops-no-bias', GATConv(16, 8, heads=2, add_self_loops=False,
                                                    bias=False))]:
        go = torch.randn(n, conv(x, ei).size(1), generator=g)
        want = _fwd_bwd(conv, (x, ei), go)
        dconv = copy.deepcopy(conv).to(dev)
        out = dconv(x.to(dev), ei.to(dev))
        assert any('GatAttend' in f for f in nodes(out)), (name, nodes(out))
        got = _fwd_bwd(dconv, (x.to(dev), ei.to(dev)), go)
        assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what=f'{name} out')
        for a, b in zip(got[1] + got[2], want[1] + want[2]):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grads')
    # observable coefficients: the reference's own sequence, same numbers
    conv = GATConv(16, 8, heads=2)
    dconv = copy.deepcopy(conv).to(dev)
    ref_out, (ref_ei, ref_alpha) = conv(x, ei, return_attention_weights=True)
    out, (d_ei, alpha) = dconv(x.to(dev), ei.to(dev), return_attention_weights=True)
    assert not any('GatAttend' in f for f in nodes(out))
    assert torch.equal(d_ei.cpu(), ref_ei)
    assert_close(out, ref_out, rtol=1e-5, atol=2e-5, what='with attention weights')
    assert_close(alpha, ref_alpha, rtol=1e-5, atol=2e-5, what='attention weights')
    conv = GATConv(16, 8, heads=2, dropout=0.5).to(dev)      # training: dropout is in effect
    assert not any('GatAttend' in f for f in nodes(conv(x.to(dev), ei.to(dev))))
    conv.eval()
    assert any('GatAttend' in f for f in nodes(conv(x.to(dev).requires_grad_(True), ei.to(dev))))
    # the reference's GAT model (BASELINE config 3's class) end to end
    torch.manual_seed(9)
    model = GAT(16, 32, num_layers=3, out_channels=6, heads=4)
    go = torch.randn(n, 6, generator=g)
    want = _fwd_bwd(model, (x, ei), go)
    dmodel = copy.deepcopy(model).to(dev)
    got = _fwd_bwd(dmodel, (x.to(dev), ei.to(dev)), go)
    assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what='GAT out')
    for a, b in zip(got[1] + got[2], want[1] + want[2]):
        assert_close_scaled(a, b, tol=2e-5, what='GAT grads')


def test_reference_gcnconv_keeps_its_order_for_nonlinear_aggregations(pyg, installed, launches,
                                                                      dev):
    """`A (X W) = (A X) W` only for linear aggregations with the stock message.
    GCNConv(16, 40, aggr='max') — input narrower than output — and a subclass overriding
    `message` must transform first, as the reference does."""
    from torch_geometric.nn import GCNConv
    g = gen(19)
    n, e = 300, 4000
    x = torch.randn(n, 16, generator=g)
    ei = torch.randint(0, n, (2, e), generator=g)
    loops = (ei[0] == ei[1]).nonzero().view(-1)
    ei[1, loops] = (ei[1, loops] + 1) % n

    class Squared(GCNConv):
        def message(self, x_j, edge_weight):
            return edge_weight.view(-1, 1) * x_j * x_j

    torch.manual_seed(2)
    for name, conv in [('max', GCNConv(16, 40, aggr='max')), ('min', GCNConv(16, 40, aggr='min')),
                       ('mean', GCNConv(16, 40, aggr='mean')), ('squared', Squared(16, 40))]:
        go = torch.randn(n, 40, generator=g)
        want = _fwd_bwd(conv, (x, ei), go)
        got = _fwd_bwd(copy.deepcopy(conv).to(dev), (x.to(dev), ei.to(dev)), go)
        assert_close(got[0], want[0], rtol=1e-5, atol=2e-5, what=f'{name} out')
        for a, b in zip(got[1] + got[2], want[1] + want[2]):
            assert_close_scaled(a, b, tol=2e-5, what=f'{name} grads')
    # the same guard on this package's own layer
    import pytorch_geometric_amd.nn as own
    torch.manual_seed(2)
    ref = GCNConv(16, 40, aggr='max')
    mine = own.GCNConv(16, 40, aggr='max').to(dev)
    mine.load_state_dict(ref.state_dict())
    assert_close(mine(x.to(dev), ei.to(dev)), ref(x, ei).detach(), rtol=1e-5, atol=2e-5,
                 what='own GCNConv aggr=max')


def test_layers_built_before_install_keep_working(pyg, dev):
    """A layer object constructed BEFORE install() has left the generated `propagate(self,
    edge_index, x, ..., size=None)` on its class (propagate.jinja:19); the wrapper must hand
    `size` over by keyword."""
    from pytorch_geometric_amd import backend
    from torch_geometric.nn import GATConv, SAGEConv
    backend.uninstall()
    g = gen(3)
    x = torch.randn(50, 8, generator=g)
    ei = torch.randint(0, 50, (2, 300), generator=g)
    torch.manual_seed(0)
    convs = [SAGEConv(8, 4, aggr='max'), GATConv(8, 4, heads=2, dropout=0.0)]
    refs = [c(x, ei).detach() for c in convs]
    backend.install()
    try:
        for c, r in zip(convs, refs):
            assert_close(c(x, ei), r, rtol=1e-6, atol=1e-6, what='CPU after install')
            assert_close(copy.deepcopy(c).to(dev)(x.to(dev), ei.to(dev)), r, rtol=1e-5,
                         atol=2e-5, what='device after install')
    finally:
        backend.uninstall()

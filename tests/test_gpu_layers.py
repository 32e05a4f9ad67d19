"""SAGEConv / GCNConv / GATConv and the BasicGNN models on the HIP path vs the reference's golden
outputs and gradients (tests/golden/make_golden.py), plus the reference's own self-consistency
checks (fused vs unfused, edge permutation invariance, flow reversal)."""
import pytest
import torch

from tests._util import assert_close, assert_close_scaled, gen

pytestmark = pytest.mark.gpu


def _run_layer(conv, case, x, dev, *args, **kwargs):
    conv.load_state_dict(case['state'])
    conv = conv.to(dev).eval()
    xx = x.to(dev).requires_grad_(True)
    out = conv(xx, *[a.to(dev) if torch.is_tensor(a) else a for a in args], **kwargs)
    params = list(conv.parameters())
    grads = torch.autograd.grad(out, [xx] + params, case['grad_out'].to(dev), allow_unused=True)
    assert_close(out, case['out'], what='out')
    assert_close(grads[0], case['grad_x'], what='grad_x')
    for (n, _), g in zip(conv.named_parameters(), grads[1:]):
        ref = case['grad_params'][n]
        if ref is not None:
            assert_close(g, ref, atol=5e-5, rtol=5e-5, what=f'grad {n}')
    return conv


@pytest.mark.parametrize('fuse', [True, False])
def test_sage_conv_golden(dev, golden, fuse):
    from pytorch_geometric_amd.nn import SAGEConv
    gr, L = golden['graph'], golden['layers']
    for name, kw in [('sage_mean', dict(aggr='mean')), ('sage_max', dict(aggr='max')),
                     ('sage_sum_noroot', dict(aggr='sum', root_weight=False, bias=False))]:
        conv = SAGEConv(16, 24, **kw)
        conv.fuse = fuse
        _run_layer(conv, L[name], gr['x'], dev, gr['edge_index'])


@pytest.mark.parametrize('fuse', [True, False])
def test_gcn_conv_golden(dev, golden, fuse):
    from pytorch_geometric_amd.nn import GCNConv
    gr, L = golden['graph'], golden['layers']
    for name, kw, args in [('gcn', {}, (gr['edge_index'], )),
                           ('gcn_weighted', {}, (gr['edge_index'], gr['edge_weight'])),
                           ('gcn_nonorm', dict(normalize=False),
                            (gr['edge_index'], gr['edge_weight']))]:
        conv = GCNConv(16, 12, **kw)
        conv.fuse = fuse
        _run_layer(conv, L[name], gr['x'], dev, *args)


def test_gcn_conv_aggregates_at_the_narrower_width(dev, monkeypatch):
    """A (X W) = (A X) W: a GCNConv whose input is narrower than its output aggregates first (the
    reference transforms first, gcn_conv.py:260-264) — same values, and no transposed aggregation
    in the backward when ``x`` takes no gradient; ``aggregate_first = False`` keeps the order."""
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.nn import GCNConv
    from tests._util import random_graph
    g = gen(21)
    n = 2000
    ei = random_graph(n, n, 24000, seed=21).to(dev)
    x = torch.randn(n, 12, generator=g).to(dev)
    go = torch.randn(n, 40, generator=g).to(dev)
    torch.manual_seed(5)
    conv = GCNConv(12, 40).to(dev)
    sink = []
    monkeypatch.setattr(_native, 'timing_sink', sink)

    def run(first, x_grad):
        conv.aggregate_first = first
        conv.zero_grad()
        xg = x.clone().requires_grad_(x_grad)
        del sink[:]
        out = conv(xg, ei)
        out.backward(go)
        widths = sorted(i['F'] for i, _, _ in sink if 'F' in i)
        return out.detach(), xg.grad, [p.grad.clone() for p in conv.parameters()], widths

    a = run(True, True)
    b = run(False, True)
    assert a[3] == [12, 12] and b[3] == [40, 40], (a[3], b[3])
    assert_close_scaled(a[0], b[0].cpu(), what='out')
    assert_close_scaled(a[1], b[1].cpu(), what='grad x')
    for p, q in zip(a[2], b[2]):
        assert_close_scaled(p, q.cpu(), what='grad param')
    c = run(True, False)
    assert c[3] == [12], c[3]          # forward only: nothing to aggregate on the way back
    for p, q in zip(c[2], b[2]):
        assert_close_scaled(p, q.cpu(), what='grad param (x without gradient)')
    assert run(False, False)[3] == [40, 40]
    wide = GCNConv(40, 12).to(dev)      # the input is the wider side: the reference's order
    del sink[:]
    wide(torch.randn(n, 40, device=dev), ei)
    assert [i['F'] for i, _, _ in sink if 'F' in i] == [12]


def test_gcn_norm_golden(dev, golden):
    from pytorch_geometric_amd.nn import gcn_norm
    gr, lp = golden['graph'], golden['loops']
    ei, ew = gcn_norm(gr['edge_index'].to(dev), gr['edge_weight'].to(dev), gr['N'])
    assert_close(ei, lp['norm_ei'])
    assert_close(ew, lp['norm_ew'])
    ei, ew = gcn_norm(gr['edge_index'].to(dev), None, gr['N'])
    assert_close(ei, lp['norm0_ei'])
    assert_close(ew, lp['norm0_ew'])


@pytest.mark.parametrize('fuse', [True, False])
def test_gat_conv_golden(dev, golden, fuse):
    from pytorch_geometric_amd.nn import GATConv
    gr, L = golden['graph'], golden['layers']
    for name, kw in [('gat', dict(heads=4)), ('gat_mean_heads', dict(heads=4, concat=False)),
                     ('gat_noloops', dict(heads=2, add_self_loops=False))]:
        conv = GATConv(16, 6, **kw)
        conv.fuse = fuse
        _run_layer(conv, L[name], gr['x'], dev, gr['edge_index'])


@pytest.mark.parametrize('fuse', [True, False])
def test_gat_attention_weights(dev, golden, fuse):
    from pytorch_geometric_amd.nn import GATConv
    gr, ga = golden['graph'], golden['gat_attention']
    conv = GATConv(16, 6, heads=4)
    conv.load_state_dict(ga['state'])
    conv = conv.to(dev).eval()
    conv.fuse = fuse
    out, (ei, alpha) = conv(gr['x'].to(dev), gr['edge_index'].to(dev),
                            return_attention_weights=True)
    assert_close(out, ga['out'])
    assert_close(ei, ga['edge_index'])
    assert_close(alpha, ga['alpha'])


def _run_model(model, case, golden, dev):
    gr = golden['graph']
    model.load_state_dict(case['state'])
    model = model.to(dev).eval()
    xx = gr['x'].to(dev).requires_grad_(True)
    out = model(xx, gr['edge_index'].to(dev))
    params = list(model.parameters())
    grads = torch.autograd.grad(out, [xx] + params, case['grad_out'].to(dev))
    assert_close(out, case['out'], atol=2e-5, what='model out')
    assert_close(grads[0], case['grad_x'], atol=2e-5, what='model grad_x')
    for (n, _), g in zip(model.named_parameters(), grads[1:]):
        assert_close(g, case['grad_params'][n], atol=1e-4, rtol=1e-4, what=f'grad {n}')


def test_models_golden(dev, golden):
    from pytorch_geometric_amd.nn import GAT, GCN, GraphSAGE
    M = golden['models']
    _run_model(GraphSAGE(16, 32, num_layers=3, out_channels=8), M['graphsage'], golden, dev)
    _run_model(GCN(16, 16, num_layers=2, out_channels=7), M['gcn'], golden, dev)
    _run_model(GAT(16, 32, num_layers=3, out_channels=5, heads=4), M['gat'], golden, dev)


def test_permutation_flow_and_int32(dev, golden):
    """testing/asserts.py:81-88 (edge-permutation invariance), test_gcn_conv.py:107-115 (flow),
    test_message_passing.py:686-703 (int32 edge_index)."""
    from pytorch_geometric_amd.nn import GCNConv, SAGEConv
    gr = golden['graph']
    x, ei = gr['x'].to(dev), gr['edge_index'].to(dev)
    torch.manual_seed(0)
    conv = SAGEConv(16, 8).to(dev)
    out = conv(x, ei)
    perm = torch.randperm(ei.size(1), generator=gen(0)).to(dev)
    assert_close(conv(x, ei[:, perm].contiguous()), out.cpu())
    assert_close(conv(x, ei.int()), out.cpu(), rtol=0, atol=0)
    conv1 = GCNConv(16, 8, flow='source_to_target').to(dev)
    conv2 = GCNConv(16, 8, flow='target_to_source').to(dev)
    conv2.load_state_dict(conv1.state_dict())
    assert_close(conv2(x, ei.flip(0).contiguous()), conv1(x, ei).cpu())


def test_bipartite_and_handle_input(dev):
    """SAGEConv on (x_src, x_dst) with an explicit EdgeIndex handle (test_sage_conv.py:30-45)."""
    import pytorch_geometric_amd as pga
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd.nn import SAGEConv
    g = gen(4)
    x1, x2 = torch.randn(30, 8, generator=g), torch.randn(20, 8, generator=g)
    ei = torch.stack([torch.randint(0, 30, (200, ), generator=g),
                      torch.randint(0, 20, (200, ), generator=g)])
    torch.manual_seed(1)
    conv = SAGEConv((8, 8), 12)
    ref = O.sage_conv(x1, ei, conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight, 'mean',
                      x_dst=x2)
    conv = conv.to(dev)
    out = conv((x1.to(dev), x2.to(dev)), ei.to(dev))
    assert_close(out, ref.detach())
    h = pga.EdgeIndex(ei.to(dev), (30, 20))
    assert_close(conv((x1.to(dev), x2.to(dev)), h), ref.detach())
    out = conv((x1.to(dev), None), ei.to(dev), size=(30, 20))
    ref2 = O.sage_conv(x1, ei, conv.lin_l.weight.cpu(), conv.lin_l.bias.cpu(), None, 'mean',
                       num_dst=20)
    assert_close(out, ref2.detach())


def test_sage_conv_layer_runs_as_one_kernel_node(dev, monkeypatch):
    """A model built from SAGEConv layers (not the GraphSAGE class): every eligible layer is ONE
    autograd node over the one-kernel layer (nn/models/_fused_sage.py:layer_eligible) and computes
    what propagate -> lin_l -> + lin_r(x) (sage_conv.py:118-139) computes; pairs, other
    aggregations, opted-out layers and hooked layers keep the propagate path."""
    from pytorch_geometric_amd.nn import SAGEConv
    from pytorch_geometric_amd.nn.models import _fused_sage
    from tests._util import random_graph
    g = gen(12)
    n, e = 3000, 40000
    ei = random_graph(n, n, e, seed=12).to(dev)
    x = torch.randn(n, 36, generator=g).to(dev)
    w_out = torch.randn(n, 10, generator=g).to(dev)
    torch.manual_seed(3)
    convs = [SAGEConv(36, 64).to(dev), SAGEConv(64, 64, aggr='sum', normalize=True).to(dev),
             SAGEConv(64, 10, bias=False).to(dev)]   # (the last one: narrow -> transform first)

    def run():
        for c in convs:
            c.zero_grad()
        xg = x.clone().requires_grad_(True)
        h = xg
        for c in convs[:-1]:
            h = c(h, ei).relu()
        out = convs[-1](h, ei)
        (out * w_out).sum().backward()
        return out, xg.grad, [p.grad.clone() for c in convs for p in c.parameters()]

    out, gx, gp = run()
    assert 'FusedSageStack' in out.grad_fn.name()
    monkeypatch.setattr(_fused_sage, 'LAYER_NODE', False)
    ref_out, ref_gx, ref_gp = run()
    assert 'FusedSageStack' not in ref_out.grad_fn.name()
    monkeypatch.setattr(_fused_sage, 'LAYER_NODE', True)
    assert_close_scaled(out, ref_out.detach().cpu(), what='out')
    assert_close_scaled(gx, ref_gx.cpu(), what='grad_x')
    for a, b in zip(gp, ref_gp):
        assert_close_scaled(a, b.cpu(), what='param grad')
    # who steps aside
    conv = convs[0]
    assert not _fused_sage.layer_eligible(conv, x, ei, (n, n - 1))
    assert _fused_sage.layer_eligible(conv, x, ei, (n, n))
    assert not _fused_sage.layer_eligible(conv, x.double(), ei, None)
    assert not _fused_sage.layer_eligible(SAGEConv(36, 8, aggr='max').to(dev), x, ei, None)
    assert not _fused_sage.layer_eligible(SAGEConv(36, 8, project=True).to(dev), x, ei, None)
    assert not _fused_sage.layer_eligible(SAGEConv(36, 8, root_weight=False).to(dev), x, ei, None)
    conv.fuse = False
    assert not _fused_sage.layer_eligible(conv, x, ei, None)
    conv.fuse = True
    pair = SAGEConv((36, 36), 8).to(dev)((x, x[:100]), ei[:, ei[1] < 100])
    assert 'FusedSageStack' not in pair.grad_fn.name()
    handle = conv.register_propagate_forward_pre_hook(lambda m, a: None)
    assert not _fused_sage.layer_eligible(conv, x, ei, None)
    handle.remove()
    with torch.no_grad():
        assert_close_scaled(conv(x, ei), convs[0](x.clone().requires_grad_(True), ei).detach().cpu())
    # a single-use batch handle (the loader's): the layer node from SINGLE_USE_MIN_EDGES edges on
    # (it sorts the batch by source once), the atomic backward below that
    import pytorch_geometric_amd as pga
    order = ei[1].argsort(stable=True)
    batch = pga.EdgeIndex.from_sorted_batch(ei[:, order].contiguous(), n,
                                            max_in_degree=int(ei[1].bincount().max()))
    assert batch.atomic_backward and not _fused_sage.layer_eligible(conv, x, batch, None)
    monkeypatch.setattr(_fused_sage, 'SINGLE_USE_MIN_EDGES', 1000)
    assert _fused_sage.layer_eligible(conv, x, batch, None)

    def run_batch():
        conv.zero_grad()
        xg = x.clone().requires_grad_(True)
        o = conv(xg, batch)
        (o * w_out[:, :1]).sum().backward()
        return o, xg.grad, [p.grad.clone() for p in conv.parameters()]

    got = run_batch()
    assert 'FusedSageStack' in got[0].grad_fn.name()
    monkeypatch.setattr(_fused_sage, 'LAYER_NODE', False)
    want = run_batch()
    monkeypatch.setattr(_fused_sage, 'LAYER_NODE', True)
    assert 'FusedSageStack' not in want[0].grad_fn.name()
    assert_close_scaled(got[0], want[0].detach().cpu(), what='batch out')
    assert_close_scaled(got[1], want[1].cpu(), what='batch grad_x')
    for a, b in zip(got[2], want[2]):
        assert_close_scaled(a, b.cpu(), what='batch param grad')
    # widths that are no multiple of four, and GraphConv (the same layer as lin_rel / lin_root;
    # with edge weights it keeps the weighted SpMM)
    from pytorch_geometric_amd.nn import GraphConv
    x5 = torch.randn(n, 5, generator=g).to(dev)
    torch.manual_seed(4)
    odd = [SAGEConv(5, 7).to(dev), GraphConv(7, 9).to(dev), GraphConv(9, 3, aggr='mean').to(dev)]

    def run_odd():
        for c in odd:
            c.zero_grad()
        h = x5.clone().requires_grad_(True)
        first = h
        for c in odd:
            h = c(h, ei).tanh()
        h.sum().backward()
        return h, first.grad, [p.grad.clone() for c in odd for p in c.parameters()]

    got = run_odd()
    assert 'FusedSageStack' in odd[1](x5.new_zeros(n, 7), ei).grad_fn.name()
    monkeypatch.setattr(_fused_sage, 'LAYER_NODE', False)
    want = run_odd()
    monkeypatch.setattr(_fused_sage, 'LAYER_NODE', True)
    assert_close_scaled(got[0], want[0].detach().cpu(), what='odd out')
    assert_close_scaled(got[1], want[1].cpu(), what='odd grad_x')
    for a, b in zip(got[2], want[2]):
        assert_close_scaled(a, b.cpu(), what='odd param grad')
    w = torch.rand(ei.size(1), generator=g).to(dev)
    assert 'FusedSageStack' not in odd[1](x5.new_zeros(n, 7), ei, w).grad_fn.name()


def test_out_of_range_edge_index(dev):
    """test_message_passing.py:201-213: IndexError on invalid indices (unfused path)."""
    from pytorch_geometric_amd.nn import SAGEConv
    conv = SAGEConv(4, 4).to(dev)
    x = torch.randn(3, 4, device=dev)
    for fuse in (False, True):
        conv.fuse = fuse
        with pytest.raises(IndexError, match='larger than 2'):
            conv(x, torch.tensor([[0, 3], [1, 0]], device=dev))
        with pytest.raises(IndexError, match='negative indices'):
            conv(x, torch.tensor([[0, -1], [1, 0]], device=dev))
    with pytest.raises(ValueError, match='integer'):
        conv(x, torch.tensor([[0., 1.], [1., 0.]], device=dev))


def test_trim_to_layer_model(dev):
    """basic_gnn.py:229-243: hop-ordered batches only aggregate a prefix per layer."""
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd.nn import GraphSAGE
    g = gen(8)
    nodes_per_hop, edges_per_hop = [4, 10, 30], [12, 60]
    n = sum(nodes_per_hop)
    # hop h edges point from hop-(h+1) nodes to hop-h nodes
    e0 = torch.stack([torch.randint(4, 14, (12, ), generator=g),
                      torch.randint(0, 4, (12, ), generator=g)])
    e1 = torch.stack([torch.randint(14, 44, (60, ), generator=g),
                      torch.randint(4, 14, (60, ), generator=g)])
    ei = torch.cat([e0, e1], dim=1)
    x = torch.randn(n, 6, generator=g)
    torch.manual_seed(2)
    model = GraphSAGE(6, 10, num_layers=2, out_channels=3)
    st = model.state_dict()
    params = [(st[f'convs.{i}.lin_l.weight'], st[f'convs.{i}.lin_l.bias'],
               st[f'convs.{i}.lin_r.weight']) for i in range(2)]
    ref = O.graphsage(x, ei, params)[:4]
    model = model.to(dev)
    full = model(x.to(dev), ei.to(dev))[:4]
    trimmed = model(x.to(dev), ei.to(dev), num_sampled_nodes_per_hop=nodes_per_hop,
                    num_sampled_edges_per_hop=edges_per_hop)[:4]
    assert_close(full, ref.detach())
    assert_close(trimmed, ref.detach())


def test_fused_sage_stack_matches_layer_loop_and_oracle(dev):
    """The whole-stack fusion (one GEMM per layer on [agg | x], accumulate-SpMM backward) against
    the layer-by-layer path and the oracle, values and every gradient."""
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd.nn import GraphSAGE
    from tests._util import random_graph
    g = gen(21)
    n = 500
    ei = random_graph(n, n, 9000, seed=21, skew=True)
    x = torch.randn(n, 20, generator=g)
    go = torch.randn(n, 7, generator=g)
    torch.manual_seed(5)
    model = GraphSAGE(20, 32, num_layers=3, out_channels=7)
    st = model.state_dict()
    params = [(st[f'convs.{i}.lin_l.weight'].clone().requires_grad_(True),
               st[f'convs.{i}.lin_l.bias'].clone().requires_grad_(True),
               st[f'convs.{i}.lin_r.weight'].clone().requires_grad_(True)) for i in range(3)]
    xr = x.clone().requires_grad_(True)
    ref = O.graphsage(xr, ei, params)
    ref.backward(go)
    model = model.to(dev)
    results = {}
    for fused in (True, 'reference-order', False):
        model.fuse_stack = bool(fused)
        model.reorder_narrow_layers = fused is True  # 32 -> 7 output layer: transform first
        model.zero_grad()
        xg = x.to(dev).requires_grad_(True)
        out = model(xg, ei.to(dev))
        out.backward(go.to(dev))
        results[fused] = (out.detach().cpu(), xg.grad.cpu(),
                          [p.grad.detach().cpu().clone() for p in model.parameters()])
    for fused in (True, 'reference-order', False):
        out, gx, gp = results[fused]
        assert_close(out, ref.detach(), atol=2e-5, what=f'fused={fused} out')
        assert_close(gx, xr.grad, atol=2e-5, what=f'fused={fused} grad_x')
        flat_ref = [t.grad for layer in params for t in layer]
        for got, want in zip(gp, flat_ref):
            assert_close(got, want, atol=1e-4, rtol=1e-4, what=f'fused={fused} param grad')
    # x without grad (the bench configuration): no layer-0 input gradient is computed
    model.fuse_stack = True
    out = model(x.to(dev), ei.to(dev))
    out.backward(go.to(dev))
    assert_close(out, ref.detach(), atol=2e-5)


def test_fused_sage_stack_takes_an_expanded_gradient(dev):
    """A graph-level readout (`out.sum(0)`, `out.mean(0)`) hands the stack's backward a gradient
    with strides (0, 1): every row is the same memory.  The output layer here is transform-first
    ('pre'), whose backward packs the gradient rows (rows_pack) — it must see real rows (ADVICE r3:
    the expanded tensor reached the C entry with a leading dimension of 0)."""
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd.nn import GraphSAGE
    from tests._util import assert_close_scaled, random_graph
    g = gen(91)
    n = 700
    ei = random_graph(n, n, 9_000, seed=91, skew=True)
    x = torch.randn(n, 24, generator=g)
    torch.manual_seed(8)
    model = GraphSAGE(24, 32, num_layers=2, out_channels=5)   # 32 -> 5: transform first
    st = model.state_dict()
    params = [(st[f'convs.{i}.lin_l.weight'].clone().requires_grad_(True),
               st[f'convs.{i}.lin_l.bias'].clone().requires_grad_(True),
               st[f'convs.{i}.lin_r.weight'].clone().requires_grad_(True)) for i in range(2)]
    w = torch.randn(5, generator=g)
    for readout in ('sum', 'mean'):
        for p in (t for layer in params for t in layer):
            p.grad = None
        xr = x.clone().requires_grad_(True)
        ref = O.graphsage(xr, ei, params)
        (getattr(ref, readout)(0) * w).sum().backward()
        md = model.to(dev)
        md.zero_grad()
        xg = x.to(dev).requires_grad_(True)
        out = md(xg, ei.to(dev))
        (getattr(out, readout)(0) * w.to(dev)).sum().backward()
        assert_close_scaled(xg.grad, xr.grad, what=f'{readout} readout grad_x')
        flat_ref = [t.grad for layer in params for t in layer]
        for got, want in zip((p.grad for p in md.parameters()), flat_ref):
            assert_close_scaled(got, want, what=f'{readout} readout param grad')


@pytest.mark.parametrize('aggr', ['mean', 'sum'])
def test_fused_sage_stack_with_a_loss_on_a_training_split(dev, aggr, monkeypatch):
    """`out[train_idx]` leaves most rows of the incoming gradient zero: the backward of the
    transform-first output layer finds them (rows_pack) and its transposed aggregation skips them
    (src_bits).  Same gradients as the oracle and as the path that reads every row."""
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.nn import GraphSAGE
    from pytorch_geometric_amd.nn.models import _fused_sage
    from tests._util import assert_close_scaled, random_graph
    g = gen(33)
    n = 1500
    ei = random_graph(n, n, 30_000, seed=33, skew=True)
    x = torch.randn(n, 24, generator=g)
    train = torch.randperm(n, generator=g)[:n // 12]
    y = torch.randint(0, 7, (n, ), generator=g)
    torch.manual_seed(6)
    model = GraphSAGE(24, 32, num_layers=3, out_channels=7, aggr=aggr)
    st = model.state_dict()
    params = [(st[f'convs.{i}.lin_l.weight'].clone().requires_grad_(True),
               st[f'convs.{i}.lin_l.bias'].clone().requires_grad_(True),
               st[f'convs.{i}.lin_r.weight'].clone().requires_grad_(True)) for i in range(3)]
    xr = x.clone().requires_grad_(True)
    ref = O.graphsage(xr, ei, params, aggr=aggr)
    torch.nn.functional.cross_entropy(ref[train], y[train]).backward()
    model = model.to(dev)
    results = {}
    for sparse in (True, False):
        monkeypatch.setattr(_fused_sage, 'SPARSE_GRAD', sparse)
        # (the opt-in compressed copy of the hidden activations rides along on one of the runs)
        monkeypatch.setattr(_fused_sage, 'COMPRESS_ROWS', sparse)
        model.zero_grad()
        xg = x.to(dev).requires_grad_(True)
        sink = []
        monkeypatch.setattr(_native, 'timing_sink', sink)
        out = model(xg, ei.to(dev))
        torch.nn.functional.cross_entropy(out[train.to(dev)], y[train].to(dev)).backward()
        monkeypatch.setattr(_native, 'timing_sink', None)
        used = [info for info, *_ in sink if info.get('src_bits')]
        assert (len(used) == 1 and used[0]['F'] == 8) if sparse else not used
        results[sparse] = (xg.grad.cpu(), [p.grad.detach().cpu().clone()
                                           for p in model.parameters()])
    flat_ref = [t.grad for layer in params for t in layer]
    for sparse in (True, False):
        gx, gp = results[sparse]
        assert_close_scaled(gx, xr.grad, what=f'sparse={sparse} grad_x')
        for got, want in zip(gp, flat_ref):
            assert_close_scaled(got, want, what=f'sparse={sparse} param grad')


def test_rgcn_conv_golden(dev, golden):
    """config 5 (a16): per-relation mean + W_r, incl. basis and block-diagonal decompositions."""
    from pytorch_geometric_amd.nn import RGCNConv
    gr, L = golden['graph'], golden['layers']
    args = (gr['edge_index'], gr['edge_type'])
    _run_layer(RGCNConv(16, 10, num_relations=5), L['rgcn'], gr['x'], dev, *args)
    _run_layer(RGCNConv(16, 12, num_relations=5, num_blocks=4), L['rgcn_blocks'], gr['x'], dev,
               *args)
    _run_layer(RGCNConv(16, 10, num_relations=5, num_bases=3), L['rgcn_bases'], gr['x'], dev,
               *args)


def test_fast_rgcn_and_index_inputs_golden(dev, golden, golden_rgcn):
    """FastRGCNConv and the node-index inputs of both classes against the real reference."""
    from pytorch_geometric_amd.nn import FastRGCNConv, RGCNConv
    gr, L = golden['graph'], golden_rgcn['layers']
    args = (gr['edge_index'], gr['edge_type'])
    N = gr['x'].size(0)
    _run_layer(FastRGCNConv(16, 10, num_relations=5), L['fast'], gr['x'], dev, *args)
    _run_layer(FastRGCNConv(16, 10, num_relations=5, aggr='add'), L['fast_add'], gr['x'], dev,
               *args)
    _run_layer(FastRGCNConv(16, 12, num_relations=5, num_blocks=4), L['fast_blocks'], gr['x'],
               dev, *args)
    _run_layer(FastRGCNConv(16, 10, num_relations=5, num_bases=3), L['fast_bases'], gr['x'], dev,
               *args)

    def run_index(conv, case, x):
        conv.load_state_dict(case['state'])
        conv = conv.to(dev).eval()
        out = conv(None if x is None else x.to(dev), *[a.to(dev) for a in args])
        params = list(conv.parameters())
        grads = torch.autograd.grad(out, params, case['grad_out'].to(dev), allow_unused=True)
        assert_close(out, case['out'], what='out')
        for (n, _), g in zip(conv.named_parameters(), grads):
            if case['grad_params'][n] is not None:
                assert_close(g, case['grad_params'][n], atol=5e-5, rtol=5e-5, what=f'grad {n}')

    x_idx = golden_rgcn['x_idx']
    run_index(RGCNConv(16, 10, num_relations=5), L['rgcn_index'], x_idx)
    run_index(RGCNConv(16, 10, num_relations=5, aggr='max'), L['rgcn_index_max'], x_idx)
    run_index(RGCNConv(N, 10, num_relations=5), L['rgcn_none'], None)
    run_index(FastRGCNConv(N, 10, num_relations=5), L['fast_none'], None)
    run_index(FastRGCNConv(N, 10, num_relations=5, num_bases=3), L['fast_none_bases'], None)
    with pytest.raises(AssertionError):
        FastRGCNConv(16, 10, num_relations=5, aggr='max')(gr['x'].to(dev),
                                                          *[a.to(dev) for a in args])


def test_rgcn_conv_vs_oracle_skewed_relations(dev):
    """FB15k-237-like relation histogram (Zipf), empty relations, int32 indices, sum aggregation."""
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd.nn import RGCNConv
    from tests._util import assert_close_scaled, random_graph
    g = gen(33)
    n, e, R = 300, 6000, 40
    ei = random_graph(n, n, e, seed=33, skew=True)
    et = (torch.rand(e, generator=g).pow(3) * R).long().clamp(max=R - 1)
    et[et == 7] = 8  # relation 7 is empty
    x = torch.randn(n, 12, generator=g)
    go = torch.randn(n, 9, generator=g)
    for aggr in ('mean', 'sum'):
        torch.manual_seed(3)
        conv = RGCNConv(12, 9, num_relations=R, aggr=aggr)
        w, r, b = (conv.weight.detach().clone().requires_grad_(True),
                   conv.root.detach().clone().requires_grad_(True),
                   conv.bias.detach().clone().requires_grad_(True))
        xr = x.clone().requires_grad_(True)
        ref = O.rgcn_conv(xr, ei, et, w, r, b, aggr)
        ref.backward(go)
        conv = conv.to(dev)
        xg = x.to(dev).requires_grad_(True)
        out = conv(xg, ei.int().to(dev), et.to(dev))
        out.backward(go.to(dev))
        assert_close(out, ref.detach(), atol=5e-5, what=f'rgcn {aggr}')
        assert_close(xg.grad, xr.grad, atol=5e-5, what=f'rgcn {aggr} grad_x')
        assert_close(conv.weight.grad, w.grad, atol=1e-4, rtol=1e-4, what='grad weight')
        assert_close(conv.root.grad, r.grad, atol=1e-4, rtol=1e-4, what='grad root')


def test_segment_matmul(dev):
    """pyg_lib.ops.segment_matmul contract vs HeteroLinear's naive loop (nn/dense/linear.py:248)."""
    from pytorch_geometric_amd.utils import segment_matmul
    g = gen(5)
    x = torch.randn(50, 6, generator=g)
    w = torch.randn(4, 6, 3, generator=g)
    ptr = torch.tensor([0, 10, 10, 35, 50])
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = torch.cat([xr[ptr[i]:ptr[i + 1]] @ wr[i] for i in range(4)])
    go = torch.randn(50, 3, generator=g)
    ref.backward(go)
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    out = segment_matmul(xg, ptr.to(dev), wg)
    out.backward(go.to(dev))
    assert_close(out, ref.detach())
    assert_close(xg.grad, xr.grad)
    assert_close(wg.grad, wr.grad, atol=2e-5)
    with pytest.raises(ValueError):
        segment_matmul(xg, torch.tensor([0, 50]), wg)
    # widths that need several k / n tiles, ragged segment sizes, K not a multiple of the chunk
    for K, N in [(100, 100), (500, 64), (33, 130)]:
        x = torch.randn(700, K, generator=g)
        w = torch.randn(6, K, N, generator=g) / K ** 0.5
        ptr = [0, 1, 1, 70, 200, 201, 700]
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ref = torch.cat([xr[ptr[i]:ptr[i + 1]] @ wr[i] for i in range(6)])
        go = torch.randn(700, N, generator=g)
        ref.backward(go)
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        out = segment_matmul(xg, ptr, wg)
        out.backward(go.to(dev))
        assert_close(out, ref.detach(), atol=2e-5, what=f'segmm {K}x{N}')
        assert_close(xg.grad, xr.grad, atol=2e-5, what=f'segmm grad_x {K}x{N}')
        assert_close(wg.grad, wr.grad, atol=1e-4, rtol=1e-4, what=f'segmm grad_w {K}x{N}')


def test_block_segment_matmul(dev):
    """Block-diagonal grouped GEMM (RGCNConv num_blocks, rgcn_conv.py:222-244) vs the reference's
    einsum 'abc,bcd->abd' per relation, values and both gradients."""
    from pytorch_geometric_amd.utils._segment_matmul import block_segment_matmul
    g = gen(6)
    for R, B, K, N, ptr in [(3, 2, 4, 3, (0, 5, 5, 40)), (4, 5, 100, 100, (0, 1, 300, 300, 777)),
                            (2, 3, 20, 36, (0, 0, 130))]:
        S = ptr[-1]
        x = torch.randn(S, B * K, generator=g)
        w = torch.randn(R, B, K, N, generator=g) / K ** 0.5
        go = torch.randn(S, B * N, generator=g)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ref = torch.cat([torch.einsum('abc,bcd->abd', xr[ptr[r]:ptr[r + 1]].view(-1, B, K),
                                      wr[r]).reshape(-1, B * N) for r in range(R)])
        ref.backward(go)
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        out = block_segment_matmul(xg, ptr, wg)
        out.backward(go.to(dev))
        assert_close(out, ref.detach(), atol=2e-5, what=f'block segmm {R}x{B}x{K}x{N}')
        assert_close(xg.grad, xr.grad, atol=2e-5, what='block segmm grad_x')
        assert_close(wg.grad, wr.grad, atol=1e-4, rtol=1e-4, what='block segmm grad_w')


def test_unfused_path_flow_and_sort_order_validation(dev):
    """fuse=False with flow='target_to_source' equals the fused path on the flipped edge list;
    a wrong sort_order claim is rejected."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import SAGEConv
    from tests._util import random_graph
    ei = random_graph(50, 50, 600, seed=2).to(dev)
    x = torch.randn(50, 6, generator=gen(2)).to(dev)
    torch.manual_seed(0)
    a = SAGEConv(6, 5, flow='target_to_source').to(dev)
    b = SAGEConv(6, 5).to(dev)
    b.load_state_dict(a.state_dict())
    ref = b(x, ei.flip(0).contiguous())
    a.fuse = False
    assert_close(a(x, ei), ref.cpu(), atol=2e-5)
    a.fuse = True
    assert_close(a(x, ei), ref.cpu(), atol=2e-5)
    with pytest.raises(ValueError, match='not sorted'):
        pga.EdgeIndex(ei, (50, 50), sort_order='col')


@pytest.mark.parametrize('fuse', [True, False])
def test_graph_conv_golden(dev, golden, fuse):
    from pytorch_geometric_amd.nn import GraphConv
    gr, L = golden['graph'], golden['layers']
    for name, kw, args in [('graph_add_weighted', dict(aggr='add'),
                            (gr['edge_index'], gr['edge_weight'])),
                           ('graph_mean', dict(aggr='mean'), (gr['edge_index'], )),
                           ('graph_max', dict(aggr='max'), (gr['edge_index'], ))]:
        conv = GraphConv(16, 10, **kw)
        conv.fuse = fuse
        _run_layer(conv, L[name], gr['x'], dev, *args)


def test_training_step_is_hipgraph_capturable(dev):
    """Launch-bound small graphs (config 1): after one eager step built the cached handles, a whole
    forward+backward captures into a HIP graph (our launches go to the capturing stream, outputs
    come from torch's graph-aware allocator, no host sync) and replays to the same numbers."""
    from pytorch_geometric_amd.nn import GCN
    from tests._util import random_graph
    g = gen(3)
    n = 500
    ei = random_graph(n, n, 4000, seed=3).to(dev)
    x = torch.randn(n, 32, generator=g).to(dev)
    torch.manual_seed(0)
    model = GCN(32, 16, num_layers=2, out_channels=7, cached=True).to(dev)
    static_x = x.clone()

    def step():
        model.zero_grad(set_to_none=False)
        out = model(static_x, ei)
        out.square().mean().backward()
        return out

    from pytorch_geometric_amd.hipgraph import CapturedStep
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    ref_out = step().detach().clone()  # eager reference (also builds and caches the handles)
    ref_grads = [p.grad.clone() for p in model.parameters()]
    replay = CapturedStep(step, warmup=2)
    cap_out = replay.output
    static_x.copy_(x * 2)       # new input through the static buffer
    replay()
    static_x.copy_(x)           # and back: must reproduce the eager numbers
    assert replay() is cap_out
    torch.cuda.synchronize()
    assert_close(cap_out, ref_out.cpu(), atol=1e-6, rtol=1e-6)
    for p, r in zip(model.parameters(), ref_grads):
        assert_close(p.grad, r.cpu(), atol=1e-6, rtol=1e-6)


def test_handles_are_normalised_and_self_looped_like_tensors(dev):
    """Round-1 ADVICE: GCNConv(normalize=True) / GATConv(add_self_loops=True) must treat an
    EdgeIndex handle exactly like the raw tensor (the reference's EdgeIndex IS a Tensor and takes
    the gcn_norm / self-loop branches, gcn_conv.py:241-258, gat_conv.py:334-347) — and the derived
    graph is built once per handle."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import GATConv, GCNConv
    g = gen(41)
    n = 120
    x = torch.randn(n, 10, generator=g).to(dev)
    ei = torch.randint(0, n, (2, 900), generator=g).to(dev)
    ew = torch.rand(900, generator=g).to(dev)
    torch.manual_seed(5)
    for conv, extra in ((GCNConv(10, 6).to(dev), ()), (GCNConv(10, 6, improved=True).to(dev),
                                                        (ew, )),
                        (GATConv(10, 4, heads=2).to(dev), ())):
        handle = pga.EdgeIndex(ei, (n, n))
        want = conv(x, ei, *extra)
        got = conv(x, handle, *extra)
        # (degrees come from an atomic scatter-add: last-ulp differences between two evaluations)
        assert_close(got, want, rtol=1e-6, atol=1e-6, what=type(conv).__name__)
        if not extra:
            assert len(handle._derived) == 1
            first = next(iter(handle._derived.values()))[0]
            conv(x, handle)
            assert next(iter(handle._derived.values()))[0] is first  # no rebuild, no re-sort
    with pytest.raises(ValueError, match='square graph'):
        GCNConv(10, 6).to(dev)(x, pga.EdgeIndex(ei, (n, n + 3)))


def test_gat_returns_post_dropout_attention(dev):
    """return_attention_weights gives what edge_update returns in the reference: the coefficients
    AFTER dropout, on the fused and on the unfused path alike."""
    from pytorch_geometric_amd.nn import GATConv
    g = gen(43)
    n = 200
    x = torch.randn(n, 8, generator=g).to(dev)
    ei = torch.randint(0, n, (2, 3000), generator=g).to(dev)
    for fuse in (True, False):
        conv = GATConv(8, 4, heads=2, dropout=0.5).to(dev).train()
        conv.fuse = fuse
        _, (coo, alpha) = conv(x, ei, return_attention_weights=True)
        frac = float((alpha == 0).float().mean())
        assert 0.4 < frac < 0.6, (fuse, frac)       # half of the coefficients were dropped
        kept = alpha[alpha != 0]
        assert float(kept.min()) > 0 and coo.size(1) == alpha.size(0)
        conv.eval()
        _, (_, alpha) = conv(x, ei, return_attention_weights=True)
        assert float((alpha == 0).float().mean()) < 0.01


def test_key_range_is_validated_before_limited_bit_sorts(dev):
    """Round-1 ADVICE: index_sort only sorts the bits `max_value` needs; layers that derive
    max_value from a constructor argument must reject keys beyond it instead of mis-sorting."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import HeteroLinear, RGCNConv
    g = gen(44)
    x = torch.randn(30, 6, generator=g).to(dev)
    ei = torch.randint(0, 30, (2, 100), generator=g).to(dev)
    et = torch.randint(0, 4, (100, ), generator=g).to(dev)
    conv = RGCNConv(6, 5, num_relations=4).to(dev)
    conv(x, ei, et)
    et_bad = et.clone()
    et_bad[7] = 4
    with pytest.raises(IndexError, match=r"'edge_type' must lie in \[0, 4\)"):
        conv(x, ei, et_bad)
    lin = HeteroLinear(6, 5, num_types=3).to(dev)
    tv = torch.randint(0, 3, (30, ), generator=g).to(dev)
    lin(x, tv)
    tv[3] = 3
    with pytest.raises(IndexError, match=r"'type_vec' must lie in \[0, 3\)"):
        lin(x, tv)
    # sparse min/max with non-unit values: refused, not silently un-weighted
    adj = torch.sparse_coo_tensor(ei, torch.rand(100, generator=g).to(dev), (30, 30)).coalesce()
    with pytest.raises(NotImplementedError, match='non-unit values'):
        pga.utils.spmm(adj.to_sparse_csr(), x, 'max')
    ones = torch.sparse_coo_tensor(adj.indices(), torch.ones_like(adj.values()), (30, 30))
    out = pga.utils.spmm(ones.coalesce().to_sparse_csr(), x, 'max')
    assert out.shape == (30, 6)


@pytest.mark.parametrize('kind', ['gcn', 'gat'])
def test_basic_gnn_applies_bias_and_relu_in_the_layer(dev, kind):
    """ReLU stacks of GCNConv / GATConv: the layer applies `out + bias` and the model's ReLU in ONE
    pass (_functions.BiasActFunction; backward = masked gradient + bias gradient from one read)
    instead of an ATen add in the layer and an ATen clamp in the model.  Same values and gradients
    as the unfused evaluation (LeakyReLU(0) is not a ReLU instance: no fusion), and a layer called
    on its own afterwards still returns the reference's pre-activation output.  The request is
    call-time state (a thread-local slot consumed by the layer call, nn/conv/_act_request.py):
    nothing is written on the modules, and a layer with a forward hook is not asked to fuse — its
    hook sees the pre-activation output."""
    from pytorch_geometric_amd.nn import GAT, GCN
    from tests._util import assert_close_scaled, random_graph
    g = gen(5)
    n = 900
    ei = random_graph(n, n, 12_000, seed=6).to(dev)
    x = torch.randn(n, 24, generator=g).to(dev)
    go = torch.randn(n, 6, generator=g).to(dev)
    torch.manual_seed(3)
    model = (GCN(24, 32, num_layers=3, out_channels=6) if kind == 'gcn'
             else GAT(24, 32, num_layers=3, out_channels=6, heads=4)).to(dev)
    for conv in model.convs:  # (zero-initialised biases would hide a dropped bias)
        torch.nn.init.normal_(conv.bias, std=0.5)
    res = []
    for act in (torch.nn.ReLU(), torch.nn.LeakyReLU(0.0)):
        model.act = act
        model.zero_grad()
        xg = x.clone().requires_grad_(True)
        out = model(xg, ei)
        out.backward(go)
        res.append((out.detach(), xg.grad, [p.grad.clone() for p in model.parameters()]))
        assert all(not hasattr(c, 'fused_act') for c in model.convs)
    assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-5, what='fused bias+ReLU output')
    assert_close_scaled(res[0][1], res[1][1], what='fused bias+ReLU grad_x')
    for a, b in zip(res[0][2], res[1][2]):
        assert_close_scaled(a, b, what='fused bias+ReLU parameter gradient')
    alone = model.convs[0](x, ei)
    assert bool((alone < 0).any()), 'a layer called on its own must not apply the ReLU'
    # which path ran: the one-pass kernel is asked for ReLU by the two inner layers of a ReLU stack
    from pytorch_geometric_amd import _functions
    from pytorch_geometric_amd.nn.conv import _act_request
    asked = []
    real = _functions.BiasActFunction.apply

    class Spy:
        @staticmethod
        def apply(out, bias, relu):
            asked.append(bool(relu))
            return real(out, bias, relu)

    _functions.BiasActFunction, keep = Spy, _functions.BiasActFunction
    # (the spy sits on the Python node: keep the C++ autograd nodes out of this part)
    keep_cpp, _functions.CPP_AUTOGRAD = _functions.CPP_AUTOGRAD, False
    try:
        model.act = torch.nn.ReLU()
        model(x, ei)
        assert asked == [True, True, False], asked
        # a forward hook on a layer observes its output: that layer is not asked to fuse, and the
        # hook sees negative (pre-activation) values; the model's result is unchanged
        seen = []
        hook = model.convs[0].register_forward_hook(lambda m, i, o: seen.append(o.detach()))
        del asked[:]
        hooked = model(x, ei)
        hook.remove()
        assert asked == [False, True, False], asked
        assert bool((seen[0] < 0).any())
        assert_close(hooked.detach(), res[0][0], rtol=1e-5, atol=1e-5, what='hooked stack output')
        # a request never outlives the layer call it was made for, and is this thread's alone
        import threading
        with _act_request.request_activation(model.convs[0], 'relu'):
            other = []
            t = threading.Thread(
                target=lambda: other.append(_act_request.requested_activation(model.convs[0])))
            t.start()
            t.join()
            assert other == [None]
            assert _act_request.requested_activation(model.convs[1]) is None
            assert _act_request.requested_activation(model.convs[0]) == 'relu'
            assert _act_request.requested_activation(model.convs[0]) is None  # consumed
        assert _act_request.requested_activation(model.convs[0]) is None
    finally:
        _functions.BiasActFunction = keep
        _functions.CPP_AUTOGRAD = keep_cpp
    # the one-pass kernel on a block larger than its capped grid (grid-stride loop), strided input
    from pytorch_geometric_amd import _native
    big = torch.randn(300_000, 72, generator=g).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    got = _native.bias_act(big[:, 4:68], b, True)
    assert torch.equal(got, (big[:, 4:68] + b).relu())


def test_gat_single_autograd_node_matches_the_three_function_path(dev):
    """`GATConv` runs node terms + edge softmax + aggregation as ONE autograd node
    (`GatAttendFunction`: the two gradients of the projected features meet inside the node-term
    backward kernel) when nothing between them is observable; asking for the attention weights
    takes the three-function path.  Same kernels, same order: outputs bitwise equal, gradients
    equal up to the one reordered addition."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import GATConv
    g = gen(21)
    n, e = 500, 6000
    x = torch.randn(n, 24, generator=g).to(dev)
    ei = torch.randint(0, n, (2, e), generator=g).to(dev)
    go = torch.randn(n, 4 * 8, generator=g).to(dev)
    torch.manual_seed(0)
    conv = GATConv(24, 8, heads=4).to(dev)

    def run(**kw):
        for p in conv.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        out = conv(xx, ei, **kw)
        out = out[0] if isinstance(out, tuple) else out
        out.backward(go)
        return out.detach(), xx.grad, [p.grad.clone() for p in conv.parameters()]

    one = run()
    three = run(return_attention_weights=True)
    assert torch.equal(one[0], three[0])
    assert_close(one[1], three[1], rtol=1e-5, atol=1e-6, what='grad x')
    for a, b in zip(one[2], three[2]):
        assert_close_scaled(a, b, tol=2e-6, what='grad param')
    # a handle as input and eval mode take the same node
    h = pga.EdgeIndex(ei, (n, n))
    conv.eval()
    assert torch.equal(conv(x, h), conv(x, ei))


def test_gat_takes_half_precision_and_autocast_inputs(dev):
    """the GAT kernels (the one-node attention `GatAttendFunction`, HeadDot, the edge
    softmax) compute in float32.  bf16 features — a bf16 model, or an autocast region whose Linear
    ran in bf16 — are widened at the layer's entry and take the SAME native path (they used to
    reach float32-only kernels and raise); outside autocast the result comes back in the input
    dtype.  A single-head layer on a handle marked `atomic_backward` keeps the route whose
    backward does not sort by source."""
    from pytorch_geometric_amd import _functions
    from pytorch_geometric_amd.edge_index import EdgeIndex
    from pytorch_geometric_amd.nn import GATConv
    g = gen(12)
    n, e = 400, 5000
    x = torch.randn(n, 16, generator=g).to(dev)
    ei = torch.randint(0, n, (2, e), generator=g).to(dev)
    torch.manual_seed(0)
    conv = GATConv(16, 8, heads=2).to(dev)
    want = conv(x, ei).detach()
    used = []
    real = _functions.GatAttendFunction.apply

    class Spy:
        @staticmethod
        def apply(*a):
            used.append(a[0].dtype)
            return real(*a)

    import pytorch_geometric_amd.nn.conv.gat_conv as gat_mod
    keep, gat_mod.GatAttendFunction = gat_mod.GatAttendFunction, Spy
    try:
        conv(x, ei)
        assert used == [torch.float32]
        xg = x.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = conv(xg, ei)
        assert used == [torch.float32] * 2, 'the fused node sees widened (float32) features'
        assert_close(out.float(), want, rtol=5e-2, atol=5e-2, what='GAT under autocast')
        out.float().sum().backward()
        assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())
        half = GATConv(16, 8, heads=2).to(dev).to(torch.bfloat16)
        half.load_state_dict({k: v.to(torch.bfloat16) for k, v in conv.state_dict().items()})
        out = half(x.to(torch.bfloat16), ei)
        assert out.dtype == torch.bfloat16 and used == [torch.float32] * 3
        assert_close(out.float(), want, rtol=1e-1, atol=1e-1, what='bf16 GAT')
        # one-shot handle, one head: HeadDot + SpmmFunction (atomic backward), no by-source sort
        torch.manual_seed(1)
        one = GATConv(16, 8, heads=1, add_self_loops=False).to(dev)
        h = EdgeIndex(ei, (n, n), validate=False)
        ref_out = one(x, h)
        assert len(used) == 4
        h2 = EdgeIndex(ei, (n, n), validate=False)
        h2.atomic_backward = True
        xg = x.clone().requires_grad_(True)
        out = one(xg, h2)
        assert len(used) == 4
        assert_close(out, ref_out.detach(), rtol=1e-5, atol=1e-5, what='one-shot GAT output')
        out.sum().backward()
        assert h2._csc is None, 'the one-shot route must not build the by-source form'
    finally:
        gat_mod.GatAttendFunction = keep


@pytest.mark.parametrize('C', [47, 130, 300])   # one / three register slots per lane; the > 256 loop
@pytest.mark.parametrize('with_index', [True, False])
def test_rows_cross_entropy_equals_torch(dev, with_index, C):
    """nn.functional.cross_entropy(out, y, index) — the train-split loss of a full-batch model as
    one pass over the selected rows — against F.cross_entropy(out[index], y[index]): value,
    gradient (an upstream factor, duplicate indices, strided logits), the out-of-range flag."""
    import torch.nn.functional as F

    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn.functional import cross_entropy
    g = gen(12)
    N = 5000
    wide = (torch.randn(N, C + 9, generator=g) * 2).to(dev)
    y = torch.randint(0, C, (N, ), generator=g).to(dev)
    idx = None
    if with_index:
        idx = torch.randperm(N, generator=g)[:777]
        idx = torch.cat([idx, idx[:5]]).to(dev)            # a few duplicates
    for strided in (False, True):   # plain logits, and a row-strided view of a wider buffer
        base = wide if strided else wide[:, :C].contiguous()
        x1 = base.detach().clone().requires_grad_(True)
        x2 = base.detach().clone().requires_grad_(True)
        l1, l2 = (x1[:, :C], x2[:, :C]) if strided else (x1, x2)
        got = cross_entropy(l1, y, idx)
        ref = F.cross_entropy(l2 if idx is None else l2[idx], y if idx is None else y[idx])
        r = float(ref.detach())
        assert abs(float(got.detach()) - r) <= 1e-6 * max(1.0, abs(r)), (float(got.detach()), r)
        (got * 1.7).backward()
        (ref * 1.7).backward()
        assert_close(x1.grad, x2.grad, rtol=1e-5, atol=1e-8, what='rows cross entropy: gradient')
    pga.check_index_errors()
    bad = y.clone()
    bad[3 if idx is None else int(idx[0])] = C + 4
    cross_entropy(wide[:, :C].contiguous(), bad, idx)
    with pytest.raises(IndexError):
        pga.check_index_errors()

"""Host-side logic that needs no GPU: argument validation of the dispatchers, module structure
and initialisers, the no-CPU-fallback rule, synthetic data generators."""
import math

import pytest
import torch


def test_no_cpu_fallback():
    import pytorch_geometric_amd as pga
    x = torch.randn(4, 3)
    idx = torch.tensor([0, 1, 0, 1])
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        pga.utils.scatter(x, idx, dim_size=2)
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        pga.utils.index_sort(idx)
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        pga.nn.SAGEConv(3, 2)(x, torch.tensor([[0, 1], [1, 0]]))
    ei = torch.tensor([[0, 1, 1], [1, 0, 0]])
    for fn in (pga.utils.sort_edge_index, pga.utils.coalesce, pga.utils.to_undirected):
        with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
            fn(ei, num_nodes=2)
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        pga.utils.segment_matmul(torch.randn(4, 3), [0, 4], torch.randn(1, 3, 2))


def test_layer_nodes_step_aside_off_the_device():
    """The one-kernel layer route of SAGEConv / GraphConv / the aggregate-first order of GCNConv
    are taken for float32 DEVICE features only: on CPU tensors the layers keep their general path,
    which raises (no CPU fallback) — nothing is computed on the host."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn.models import _fused_sage
    x = torch.randn(4, 3)
    ei = torch.tensor([[0, 1, 2], [1, 0, 3]])
    for conv in (pga.nn.SAGEConv(3, 8), pga.nn.GraphConv(3, 8)):
        assert not _fused_sage.layer_eligible(conv, x, ei, None)
        with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
            conv(x, ei)
    gcn = pga.nn.GCNConv(3, 8)
    assert not gcn._aggregate_first(x)
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        gcn(x, ei)


def test_product_never_imports_the_oracle():
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'pytorch_geometric_amd')
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f


def test_dispatcher_validation_matches_reference_messages():
    import pytorch_geometric_amd as pga
    src = torch.randn(2, 5, 2)
    idx = torch.tensor([0, 1, 0, 1, 0])
    with pytest.raises(ValueError, match='must be one-dimensional'):
        pga.utils.scatter(src, idx.view(1, -1))
    with pytest.raises(ValueError, match='must lay between 0 and 2'):
        pga.utils.scatter(src, idx, dim=3)
    with pytest.raises(ValueError, match="invalid `reduce` argument 'std'"):
        pga.utils.scatter(src, idx, dim=1, dim_size=2, reduce='std')
    with pytest.raises(NotImplementedError, match='float32 only'):
        pga.utils.scatter(src.double(), idx, dim=1, dim_size=2)
    with pytest.raises(ValueError, match='shape'):
        pga.EdgeIndex(torch.zeros(3, 4, dtype=torch.long))
    with pytest.raises(ValueError, match='unsupported data type'):
        pga.EdgeIndex(torch.zeros(2, 4))
    conv = pga.nn.SAGEConv(4, 4)
    with pytest.raises(ValueError, match='integer'):
        conv(torch.randn(3, 4), torch.zeros(2, 2))
    with pytest.raises(ValueError, match='two-dimensional'):
        conv(torch.randn(3, 4), torch.zeros(2, dtype=torch.long))
    with pytest.raises(ValueError, match="size '2'"):
        conv(torch.randn(3, 4), torch.zeros(3, 2, dtype=torch.long))
    with pytest.raises(ValueError, match='does not support bipartite'):
        pga.nn.GCNConv(4, 4)((torch.randn(3, 4), torch.randn(3, 4)),
                             torch.zeros(2, 2, dtype=torch.long))
    with pytest.raises(ValueError, match='does not support adding self-loops'):
        pga.nn.GCNConv(4, 4, add_self_loops=True, normalize=False)
    with pytest.raises(ValueError, match="'flow'"):
        pga.nn.SAGEConv(4, 4, flow='sideways')


def test_modules_match_reference_structure_and_init():
    from pytorch_geometric_amd.nn import GAT, GCN, GATConv, GCNConv, GraphSAGE, SAGEConv
    m = GraphSAGE(100, 256, num_layers=3, out_channels=47)
    assert sum(p.numel() for p in m.parameters()) == 206_895
    assert [tuple(c.lin_l.weight.shape) for c in m.convs] == [(256, 100), (256, 256), (47, 256)]
    assert sorted(SAGEConv(8, 4).state_dict()) == ['lin_l.bias', 'lin_l.weight', 'lin_r.weight']
    assert sorted(GCNConv(8, 4).state_dict()) == ['bias', 'lin.weight']
    assert sorted(GATConv(8, 4, heads=2).state_dict()) == ['att_dst', 'att_src', 'bias',
                                                          'lin.weight']
    gat = GAT(128, 256, num_layers=3, out_channels=40, heads=8)
    assert [(c.heads, c.out_channels, c.concat) for c in gat.convs] == [
        (8, 32, True), (8, 32, True), (8, 40, False)]
    gcn = GCNConv(1433, 16)
    bound = math.sqrt(6.0 / (1433 + 16))  # glorot
    assert gcn.lin.weight.abs().max() <= bound and (gcn.bias == 0).all()
    sage = SAGEConv(100, 256)
    kb = math.sqrt(6 / ((1 + 5) * 100))  # kaiming_uniform(a=sqrt(5)), fan=in
    assert sage.lin_l.weight.abs().max() <= kb
    assert sage.lin_l.bias.abs().max() <= 1 / math.sqrt(100)
    assert repr(sage) == 'SAGEConv(100, 256, aggr=mean)'
    assert repr(GCN(4, 8, 2)) == 'GCN(4, 8, num_layers=2)'


def test_self_loop_helpers_on_cpu(golden):
    """loop.py helpers are device-agnostic tensor bookkeeping; check them against the goldens."""
    from pytorch_geometric_amd.utils import add_self_loops, remove_self_loops
    gr, lp = golden['graph'], golden['loops']
    ei, _ = remove_self_loops(gr['edge_index'])
    ei, _ = add_self_loops(ei, num_nodes=gr['N'])
    assert torch.equal(ei, lp['gat_ei'])


def test_trim_to_layer():
    from pytorch_geometric_amd.utils import trim_to_layer
    x = torch.arange(10).view(10, 1)
    ei = torch.arange(14).view(1, 14).repeat(2, 1)
    x1, e1, _ = trim_to_layer(1, [2, 3, 5], [6, 8], x, ei)
    assert x1.size(0) == 5 and e1.size(1) == 6
    x0, e0, _ = trim_to_layer(0, [2, 3, 5], [6, 8], x, ei)
    assert x0 is x and e0 is ei


def test_products_like_generator_is_deterministic_and_skewed():
    from pytorch_geometric_amd.datasets import products_like
    x, y, ei, c = products_like(seed=1, scale=1 / 256)
    x2, y2, ei2, _ = products_like(seed=1, scale=1 / 256)
    assert torch.equal(ei, ei2) and torch.equal(x, x2) and c == 47
    n = x.size(0)
    assert ei.shape == (2, int(61_859_140 / 256) // 2 * 2) and x.shape == (n, 100)
    assert int(ei.max()) < n and int(ei.min()) >= 0
    half = ei.size(1) // 2
    assert torch.equal(ei[0, :half], ei[1, half:]) and torch.equal(ei[1, :half], ei[0, half:])
    deg = torch.bincount(ei[1], minlength=n)
    assert deg.max() > 20 * deg.float().mean()  # hubs exist
    _, _, ei3, _ = products_like(seed=2, scale=1 / 256)
    assert not torch.equal(ei, ei3)


def test_bench_byte_model():
    import bench
    info = dict(nnz=61_859_140, n_rows=2_449_029, n_src=2_449_029, F=256, idx_bytes=8,
                src_scale=False, weighted=False)
    b = bench.spmm_algorithmic_bytes(info)
    assert abs(b - (61_859_140 * (1024 + 8) + 2_449_030 * 8 + 2_449_029 * 1024)) < 1
    # the five SpMM passes of one step move ~291.7 GB
    total = 0
    for Fw, scaled in [(100, False), (256, False), (256, False), (256, True), (256, True)]:
        total += bench.spmm_algorithmic_bytes(dict(info, F=Fw, src_scale=scaled))
    assert 285e9 < total < 295e9


def test_propagate_hooks_run_without_a_gpu():
    """test/nn/conv/test_message_passing.py hooks: a pre-hook can replace the inputs, a post-hook
    the output; handles remove themselves."""
    import pytorch_geometric_amd as pga
    conv = pga.nn.SAGEConv(4, 4)
    seen = []

    def pre(module, inputs):
        seen.append('pre')
        raise RuntimeError('stop here')  # proves the hook ran before any kernel was needed

    h = conv.register_propagate_forward_pre_hook(pre)
    with pytest.raises(RuntimeError, match='stop here'):
        conv(torch.randn(3, 4), torch.zeros(2, 2, dtype=torch.long))
    assert seen == ['pre']
    h.remove()
    assert not conv._propagate_forward_pre_hooks
    post = conv.register_propagate_forward_hook(lambda m, i, o: o)
    assert len(conv._propagate_forward_hooks) == 1
    post.remove()


def test_edge_index_handle_tensor_protocol_on_the_host():
    """The handle is a Tensor subclass sharing its edge list's storage (edge_index.py:173 in the
    reference); no kernel is needed for the protocol itself."""
    import copy
    import pickle

    import pytorch_geometric_amd as pga
    ei = torch.tensor([[0, 1, 2, 2], [1, 2, 0, 1]])
    h = pga.EdgeIndex(ei, (3, 3), validate=False)
    assert isinstance(h, torch.Tensor) and h.data_ptr() == ei.data_ptr()
    assert h.as_tensor() is ei and h.num_edges == 4 and h.sparse_size == (3, 3)
    assert type(h[0]) is torch.Tensor and type(h.flip(0)) is torch.Tensor
    ei[0, 0] = 2  # in-place writes are seen by both, and by the version counter
    assert int(h[0, 0]) == 2 and h._version == ei._version
    assert h.to('cpu') is h and isinstance(h.to(torch.int32), pga.EdgeIndex)
    assert type(h.to(torch.float32)) is torch.Tensor
    for twin in (copy.deepcopy(h), pickle.loads(pickle.dumps(h))):
        assert isinstance(twin, pga.EdgeIndex) and twin.sparse_size == (3, 3)
        assert torch.equal(twin, ei) and twin.data_ptr() != ei.data_ptr()
    assert pga.EdgeIndex(h, (3, 3), validate=False).as_tensor() is ei  # no handle-of-handle
    assert pga.as_edge_index(h) is h


def test_rows_cross_entropy_steps_aside_for_cpu_tensors():
    """nn.functional.cross_entropy: CPU tensors (and anything else outside the one-pass kernel's
    float32 HIP logits) take F.cross_entropy on the gathered rows — no host computation of ours."""
    import torch.nn.functional as F

    from pytorch_geometric_amd.nn.functional import cross_entropy
    g = torch.Generator().manual_seed(3)
    x = torch.randn(40, 6, generator=g, requires_grad=True)
    y = torch.randint(0, 6, (40, ), generator=g)
    idx = torch.tensor([3, 9, 9, 31])
    a = cross_entropy(x, y, idx)
    b = F.cross_entropy(x[idx], y[idx])
    assert torch.equal(a, b)
    ga, = torch.autograd.grad(a, x)
    gb, = torch.autograd.grad(b, x)
    assert torch.equal(ga, gb)
    assert torch.equal(cross_entropy(x, y), F.cross_entropy(x, y))


def test_slot_trainer_needs_device_tensors():
    """slots.SlotTrainer is a device-only path: a CPU model / loader raises at construction."""
    import pytest

    from pytorch_geometric_amd.nn import GraphSAGE
    from pytorch_geometric_amd.slots import SlotTrainer

    class FakeSampler:
        replace = disjoint = False
        subgraph_type = 'directional'
        num_neighbors = [3, 2]
        colptr = torch.zeros(5, dtype=torch.int64)
        row = torch.zeros(0, dtype=torch.int64)
        seed = 0

    class FakeLoader:
        sampler = FakeSampler()
        x = torch.zeros(4, 8)
        y = torch.zeros(4, dtype=torch.int64)
        batch_size, num_nodes = 2, 4

    model = GraphSAGE(8, 16, num_layers=2, out_channels=3)
    with pytest.raises(Exception):
        SlotTrainer(model, FakeLoader())
    FakeLoader.y = None
    with pytest.raises(ValueError, match='label vector'):
        SlotTrainer(model, FakeLoader())

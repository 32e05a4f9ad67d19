"""backend.install(): rebinding mechanics inside a real torch_geometric (only available in the
build container, where the reference is mounted; skipped elsewhere).  CPU tensors must keep taking
the reference's own path — the backend steps aside, it never computes on the CPU."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import make_ref  # noqa: E402

try:  # the mounted reference (build container) or the staged copy under oracle/_ref
    pyg = make_ref.import_reference()
except ImportError:
    pytest.skip('no reference available', allow_module_level=True)


@pytest.fixture()
def installed():
    from pytorch_geometric_amd import backend
    backend.install()
    yield backend
    backend.uninstall()


def test_install_rebinds_every_importer_and_uninstall_restores(installed):
    import torch_geometric.nn.aggr.base as aggr_base
    import torch_geometric.nn.conv.gcn_conv as gcn_mod
    import torch_geometric.utils as U
    import torch_geometric.utils._softmax as sm_mod
    assert installed.is_installed()
    assert pyg.backend.mi355x is installed and pyg.backend.use_mi355x is None
    wrapped = U.scatter
    assert hasattr(wrapped, '__wrapped__')
    # modules that did `from torch_geometric.utils import scatter` see the same wrapper
    assert aggr_base.scatter is wrapped and gcn_mod.scatter is wrapped
    assert sm_mod.scatter is wrapped and sm_mod.segment is U.segment
    assert U.softmax.__wrapped__ is not U.softmax
    n_scatter = sum(1 for _, attr, old in installed._state['rebinds']
                    if old is wrapped.__wrapped__)
    assert n_scatter >= 30, n_scatter  # imported by name all over the package
    orig = wrapped.__wrapped__
    installed.uninstall()
    assert U.scatter is orig and aggr_base.scatter is orig and gcn_mod.scatter is orig
    assert not hasattr(pyg.backend, 'mi355x')
    installed.install()  # the fixture's teardown uninstalls again


def test_cpu_tensors_step_aside_to_the_reference(installed):
    from torch_geometric.nn import GATConv, GCNConv, SAGEConv
    from torch_geometric.utils import scatter, softmax
    g = torch.Generator().manual_seed(0)
    x = torch.randn(10, 8, generator=g)
    ei = torch.randint(0, 10, (2, 40), generator=g)
    idx = torch.randint(0, 4, (10, ), generator=g)
    ref = scatter.__wrapped__(x, idx, 0, 4, 'mean')
    assert torch.equal(scatter(x, idx, 0, 4, 'mean'), ref)
    assert torch.allclose(softmax(x, idx, num_nodes=4),
                          softmax.__wrapped__(x, idx, None, 4, 0))
    for conv in (SAGEConv(8, 4), GCNConv(8, 4), GATConv(8, 4, heads=2)):
        out = conv(x, ei)  # wrapped propagate -> NotImplemented -> reference path
        installed.uninstall()
        ref = conv(x, ei)
        installed.install()
        assert torch.allclose(out, ref, atol=1e-6)


def test_identity_memo_semantics():
    """install() step 5: results are keyed on tensor identity + in-place version, never values."""
    from pytorch_geometric_amd.backend import _IdentityMemo
    calls = []

    def fn(a, b, k):
        calls.append(k)
        return a + (0 if b is None else b) + k

    memo = _IdentityMemo(fn)
    a, b = torch.arange(4), torch.ones(4, dtype=torch.long)
    r1 = memo((a, b), (1, ))
    assert memo((a, b), (1, )) is r1 and calls == [1]
    assert memo((a, None), (1, )) is not r1 and calls == [1, 1]
    assert memo((a.clone(), b), (1, )) is not r1  # equal values, different tensor
    a.add_(1)  # version bump -> recompute
    r2 = memo((a, b), (1, ))
    assert r2 is not r1 and torch.equal(r2, a + b + 1)
    n = len(memo.store)
    del a
    import gc
    gc.collect()
    assert len(memo.store) < n  # entries die with their inputs
    for i in range(20):
        memo((b, None), (i, ))
    assert len(memo.store) <= 8


def test_reference_graphsage_forward_is_wrapped_and_cpu_steps_aside(installed):
    from torch_geometric.nn import GraphSAGE
    assert hasattr(GraphSAGE.forward, '__wrapped__')
    g = torch.Generator().manual_seed(1)
    x, ei = torch.randn(12, 6, generator=g), torch.randint(0, 12, (2, 40), generator=g)
    model = GraphSAGE(6, 8, num_layers=2, out_channels=3)
    out = model(x, ei)
    installed.uninstall()
    assert not hasattr(GraphSAGE.forward, '__wrapped__')
    ref = model(x, ei)
    installed.install()
    assert torch.equal(out, ref)
    from pytorch_geometric_amd.nn.models import _fused_sage
    assert not _fused_sage.eligible(model, x, ei, False)  # CPU tensors are never eligible


def test_flag_disables_the_backend(installed):
    pyg.backend.use_mi355x = False
    from pytorch_geometric_amd.backend import _enabled
    assert not _enabled()
    pyg.backend.use_mi355x = None
    assert _enabled()


# ---- the glue, with the kernels swapped for the CPU oracle ------------------------------------------
@pytest.fixture()
def oracle_kernels(monkeypatch, installed):
    """install()'s device branches cannot run against the real reference anywhere (no GPU in the
    build container, no reference on the GPU box).  Here the HIP-backed pieces UNDER the glue are
    replaced by the CPU oracle, so the glue itself — argument mapping, orientation, sizes,
    step-aside rules — is checked against the real reference's results."""
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd import _functions, backend
    from pytorch_geometric_amd import edge_index as ei_mod
    from pytorch_geometric_amd import utils as U
    calls = {'spmm': 0, 'scatter': 0, 'softmax': 0}

    class Graph:
        def __init__(self, ei, n_src, n_dst):
            self.ei, self.n_src, self.n_dst = ei, n_src, n_dst

    def as_edge_index(edge_index, num_src=None, num_dst=None, flip=False):
        assert type(edge_index) is torch.Tensor
        return Graph(edge_index.flip(0) if flip else edge_index, num_src, num_dst)

    class Spmm:
        @staticmethod
        def apply(x, weight, graph, reduce, order):
            assert order == 'coo' and x.size(0) == graph.n_src
            calls['spmm'] += 1
            msg = x[graph.ei[0]]
            if weight is not None:
                msg = msg * weight.view(*weight.shape, *([1] * (msg.dim() - weight.dim())))
            return O.scatter(msg, graph.ei[1], 0, graph.n_dst, reduce)

    def counted(name, fn):
        def run(*args, **kwargs):
            calls[name] += 1
            return fn(*args, **kwargs)
        return run

    monkeypatch.setattr(backend, '_ours',
                        lambda t: isinstance(t, torch.Tensor) and t.dtype == torch.float32)
    monkeypatch.setattr(backend, '_ours_index',
                        lambda t: type(t) is torch.Tensor and t.dim() == 2 and t.size(0) == 2
                        and t.dtype in (torch.int32, torch.int64))
    # the graph-rewrite memos call the device-side gcn_norm / self-loop helpers: covered on the
    # GPU by tests/test_gpu_reference_install.py, bypassed here
    monkeypatch.setattr(backend, '_memoisable', lambda ei, ea: False)
    monkeypatch.setattr(_functions, 'SpmmFunction', Spmm)
    monkeypatch.setattr(ei_mod, 'as_edge_index', as_edge_index)
    monkeypatch.setattr(U, 'scatter', counted('scatter', O.scatter))
    monkeypatch.setattr(U, 'softmax', counted('softmax', O.softmax))
    return calls


def _reference_result(installed, fn):
    installed.uninstall()
    try:
        return fn()
    finally:
        installed.install()


def test_fused_propagate_glue_matches_the_reference(installed, oracle_kernels):
    from torch_geometric.nn import GATConv, GCNConv, GraphConv, SAGEConv
    g = torch.Generator().manual_seed(3)
    n = 30
    x = torch.randn(n, 8, generator=g)
    ei = torch.randint(0, n, (2, 200), generator=g)
    w = torch.rand(200, generator=g)
    cases = [
        (SAGEConv(8, 5), (x, ei), {}),
        (SAGEConv(8, 5, aggr='max'), (x, ei), {}),
        (SAGEConv(8, 5, flow='target_to_source'), (x, ei), {}),
        (GCNConv(8, 5), (x, ei, w), {}),
        (GCNConv(8, 5, flow='target_to_source'), (x, ei), {}),
        (GraphConv(8, 5, aggr='mean'), (x, ei, w), {}),
        (GATConv(8, 4, heads=3), (x, ei), {}),
        (GATConv(8, 4, heads=2, concat=False), (x, ei), {}),
    ]
    for conv, args, kw in cases:
        conv.fuse_attention = False  # (GATConv: keep the propagate glue under test here; the
        before = oracle_kernels['spmm']  # one-node attention route only exists on the device)
        out = conv(*args, **kw)
        assert oracle_kernels['spmm'] == before + 1, f'{conv}: fused route not taken'
        ref = _reference_result(installed, lambda: conv(*args, **kw))
        assert torch.allclose(out, ref, atol=1e-5), conv
    # bipartite: (x_src, x_dst) with an explicit size; edges point into the 12 destination nodes
    x_dst = torch.randn(12, 8, generator=g)
    bi = torch.stack([torch.randint(0, n, (90, ), generator=g),
                      torch.randint(0, 12, (90, ), generator=g)])
    conv = SAGEConv((8, 8), 5)
    out = conv((x, x_dst), bi, size=(n, 12))
    ref = _reference_result(installed, lambda: conv((x, x_dst), bi, size=(n, 12)))
    assert out.shape == (12, 5) and torch.allclose(out, ref, atol=1e-5)
    # things the wrapper must leave to the reference: hooks, explain mode, exotic aggregations
    before = oracle_kernels['spmm']
    conv = SAGEConv(8, 5, aggr='median')
    conv(x, ei)
    conv = SAGEConv(8, 5)
    conv.register_propagate_forward_pre_hook(lambda m, inp: None)
    conv(x, ei)
    assert oracle_kernels['spmm'] == before
    # softmax / scatter calls of the reference's own code reach the rebound dispatchers
    assert oracle_kernels['softmax'] >= 2 and oracle_kernels['scatter'] >= 1


def test_edge_index_matmul_glue_matches_the_reference(installed, oracle_kernels):
    from torch_geometric import EdgeIndex
    g = torch.Generator().manual_seed(4)
    n_row, n_col = 9, 14
    raw = torch.stack([torch.randint(0, n_row, (60, ), generator=g),
                       torch.randint(0, n_col, (60, ), generator=g)])
    value = torch.rand(60, generator=g)
    for order, transpose, other_rows in (('row', False, n_col), ('col', True, n_row)):
        adj = EdgeIndex(raw, sparse_size=(n_row, n_col)).sort_by(order).values
        perm = EdgeIndex(raw, sparse_size=(n_row, n_col)).sort_by(order).indices
        val = value[perm]
        other = torch.randn(other_rows, 6, generator=g)
        for reduce in ('sum', 'mean', 'max', 'min'):
            for v in (None, val):
                if v is not None and reduce in ('max', 'min'):
                    continue
                before = oracle_kernels['spmm']
                out = adj.matmul(other, v, reduce=reduce, transpose=transpose)
                assert oracle_kernels['spmm'] == before + 1
                ref = _reference_result(
                    installed, lambda: adj.matmul(other, v, reduce=reduce, transpose=transpose))
                assert torch.allclose(out, ref, atol=1e-5), (order, reduce, v is None)
    unsorted = EdgeIndex(raw, sparse_size=(n_row, n_col))
    with pytest.raises(ValueError, match='sorted'):
        unsorted.matmul(torch.randn(n_col, 3))


def test_oracle_segment_logsumexp_matches_the_reference():
    """Pins oracle.segment_logsumexp (the checker of tests/test_gpu_ops.py::test_segment_logsumexp)
    against the real utils/_segment.py:53-80 on the CPU: dim 0 / dim 1, empty segments, grads."""
    from oracle import pyg_oracle as O
    from torch_geometric.utils import segment_logsumexp
    g = torch.Generator().manual_seed(0)
    ptr = torch.tensor([0, 0, 5, 10, 10, 15, 20])
    for shape, dim in (((20, 16), 0), ((16, 20), 1), ((20, ), 0)):
        a = torch.randn(*shape, generator=g, requires_grad=True)
        b = a.detach().clone().requires_grad_(True)
        ref, got = segment_logsumexp(a, ptr, dim), O.segment_logsumexp(b, ptr, dim)
        assert torch.allclose(got, ref, atol=1e-6)
        w = torch.randn(ref.shape, generator=g)
        (ref * w).sum().backward()
        (got * w).sum().backward()
        assert torch.allclose(b.grad, a.grad, atol=1e-6)


def test_round6_bindings_step_aside_on_the_cpu_and_are_restored(installed):
    """RGCNConv / FastRGCNConv / HeteroLinear / GATConv forwards, `MessagePassing._index_select`
    and the `pyg_lib` name of seam S2: bound by install(), the reference's own code for CPU
    tensors (also for layers built BEFORE install() and for the reference's EdgeIndex), restored
    by uninstall()."""
    import torch_geometric.nn.conv.rgcn_conv as rgcn_mod
    import torch_geometric.nn.dense.linear as lin_mod
    from torch_geometric import EdgeIndex
    from torch_geometric.nn import (FastRGCNConv, GATConv, GCNConv, HeteroLinear, MessagePassing,
                                    RGCNConv, SAGEConv)
    installed.uninstall()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(20, 8, generator=g)
    ei = torch.randint(0, 20, (2, 60), generator=g)
    et = torch.randint(0, 3, (60, ), generator=g)
    tv = torch.randint(0, 3, (20, ), generator=g)
    adj = EdgeIndex(ei, sparse_size=(20, 20)).sort_by('col').values
    mods = [(RGCNConv(8, 4, 3), (x, ei, et)), (FastRGCNConv(8, 4, 3, num_bases=2), (x, ei, et)),
            (RGCNConv(8, 4, 3, num_blocks=2), (x, ei, et)), (GATConv(8, 4, heads=2), (x, ei)),
            (HeteroLinear(8, 4, 3), (x, tv)), (GCNConv(8, 16, aggr='max'), (x, ei)),
            (SAGEConv(8, 4), (x, adj)), (GCNConv(8, 4), (x, adj)), (GATConv(8, 4), (x, adj))]
    refs = [m(*a) for m, a in mods]
    assert not hasattr(RGCNConv.forward, '__wrapped__')
    installed.install()
    for cls in (RGCNConv, FastRGCNConv, HeteroLinear, GATConv):
        assert hasattr(cls.forward, '__wrapped__'), cls
    assert hasattr(MessagePassing._index_select, '__wrapped__')
    assert rgcn_mod.pyg_lib is lin_mod.pyg_lib and hasattr(rgcn_mod.pyg_lib.ops, 'segment_matmul')
    with pytest.raises(NotImplementedError):  # no host computation behind the name
        rgcn_mod.pyg_lib.ops.segment_matmul(x, torch.tensor([0, 20]), torch.randn(1, 8, 4))
    for (m, a), r in zip(mods, refs):
        assert torch.allclose(m(*a), r, atol=1e-6), type(m).__name__
    installed.uninstall()
    assert not hasattr(RGCNConv.forward, '__wrapped__')
    assert not hasattr(MessagePassing._index_select, '__wrapped__')
    assert rgcn_mod.pyg_lib is object or not hasattr(rgcn_mod.pyg_lib, 'ops') \
        or rgcn_mod.pyg_lib.__name__ == 'pyg_lib'
    installed.install()


def test_linear_message_flow_guard():
    """aggregate-before-transform only for linear aggregations with the stock message."""
    from pytorch_geometric_amd.nn.conv.gcn_conv import GCNConv, linear_message_flow

    class Squared(GCNConv):
        def message(self, x_j, edge_weight):
            return x_j * x_j

    assert linear_message_flow(GCNConv(4, 8), GCNConv)
    assert linear_message_flow(GCNConv(4, 8, aggr='mean'), GCNConv)
    assert not linear_message_flow(GCNConv(4, 8, aggr='max'), GCNConv)
    assert not linear_message_flow(Squared(4, 8), GCNConv)
    hooked = GCNConv(4, 8)
    hooked.register_propagate_forward_pre_hook(lambda *a: None)
    assert not linear_message_flow(hooked, GCNConv)


@pytest.mark.timeout(600)
def test_reference_own_tests_pass_on_the_cpu_with_the_backend_installed():
    """The reference's OWN test modules (a fast subset; the GPU suite runs all 24 on the device,
    tests/test_gpu_reference_suite.py) with install() active and FULL_TEST on: CPU tensors step
    aside, and everything that INSPECTS the rebound names still finds the reference — TorchScript
    of functions (`torch.jit.script(softmax)`), of layers built after install()
    (`torch.jit.script(GCNConv(...))`: the generated `propagate` is rendered against the class as
    the reference left it) and of the `@overload`-ed helpers (`add_self_loops`, `coalesce`, ...),
    hooks, `decomposed_layers`, explain mode.  `GATConv`'s TorchScript case fails in the reference
    itself with this torch version and is deselected."""
    import subprocess
    root, files = make_ref.reference_tests()
    pick = [f for f in files if os.path.basename(f) in (
        'test_scatter.py', 'test_softmax.py', 'test_sort_edge_index.py', 'test_coalesce.py',
        'test_loop.py', 'test_message_passing.py', 'test_sage_conv.py', 'test_gcn_conv.py',
        'test_gat_conv.py', 'test_rgcn_conv.py', 'test_linear.py', 'test_basic.py')]
    assert len(pick) == 12
    ref_root = os.path.dirname(os.path.dirname(os.path.abspath(pyg.__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, ref_root]), FULL_TEST='1')
    cmd = [sys.executable, '-m', 'pytest', *pick, '-q', '-p', 'no:cacheprovider', '-p',
           'tests._install_plugin', '--rootdir', '/tmp', '-c', os.devnull, '-k',
           'not (test_gat_conv and not with_edge_attr and not empty_edge_index)']
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd='/tmp', timeout=550)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert ' passed' in tail and ' failed' not in tail, tail

"""The compiled PyTorch binding (csrc/torch_binding.cpp, `torch.ops.pyg_amd_c`) against the ctypes
route to the same C entry points: identical launches, so identical bits — for every operator the
binding carries, on strided operands, optional arguments and hub rows; and the package reports
which route it is on."""
import pytest
import torch

from tests._util import gen, random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture
def both_routes():
    """run(fn) -> (result on the compiled route, result on the ctypes route)"""
    from pytorch_geometric_amd import _compiled
    ns = _compiled.ops()
    assert ns is not None, _compiled.status()

    def run(fn):
        a = fn()
        _compiled._state['ns'] = None
        try:
            b = fn()
        finally:
            _compiled._state['ns'] = ns
        return a, b

    return run


def _same(a, b, what):
    if isinstance(a, (tuple, list)):
        for x, y in zip(a, b):
            _same(x, y, what)
        return
    if a is None:
        assert b is None
        return
    assert a.dtype == b.dtype and a.shape == b.shape, what
    assert torch.equal(a, b), f'{what}: compiled and ctypes routes differ'


def test_binding_is_the_default_route(dev):
    import pytorch_geometric_amd as pga
    assert pga.binding_status().startswith('compiled'), pga.binding_status()
    assert int(torch.ops.pyg_amd_c.abi_version()) == pga.load_library().pygamd_abi_version()


def test_compiled_and_ctypes_routes_launch_the_same_thing(dev, both_routes):
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    g = gen(3)
    n, F, Fo = 1500, 100, 64
    ei = random_graph(n, n, 40000, seed=5, skew=True)
    h = pga.EdgeIndex(ei.to(dev), (n, n))
    fwd, bwd = h.by_dst(), h.by_src()
    assert fwd.hub[2] > 0
    buf = torch.randn(n, 2 * F, generator=g).to(dev)
    x = buf[:, F:]                                     # row-strided operand
    w_e = torch.rand(ei.size(1), generator=g).to(dev)
    w_h = torch.rand(ei.size(1), 4, generator=g).to(dev)
    scale = fwd.inv_degree()
    for red in ('sum', 'mean', 'max'):
        _same(*both_routes(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, x, red, n_rows=n,
                                                    hub=fwd.hub, save_arg32=(red == 'max'))),
              f'spmm {red}')
    _same(*both_routes(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, x, 'sum', n_rows=n, eid=fwd.perm,
                                                w=w_e, src_scale=scale, hub=fwd.hub)), 'weighted')
    _same(*both_routes(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, x, 'sum', n_rows=n, w=w_h,
                                                hub=fwd.hub)), 'per-head weights')
    mask = torch.randn(n, F, generator=g).to(dev)
    _same(*both_routes(lambda: _native.spmm_csr(
        bwd.ptr, bwd.idx, x, 'sum', n_rows=n, hub=bwd.hub, out=torch.ones(n, F, device=dev),
        accumulate=True, relu_bits=_native.pack_relu_bits(mask))), 'accumulate + relu bits')
    rows = torch.randn(ei.size(1), 12, generator=g).to(dev)
    _same(*both_routes(lambda: _native.spmm_csr(fwd.ptr, None, rows, 'mean')), 'segment form')
    # dense transform
    w = (torch.randn(Fo, F, generator=g) * 0.1).to(dev)
    b = torch.randn(Fo, generator=g).to(dev)
    go = torch.randn(n, Fo, generator=g).to(dev)
    _same(*both_routes(lambda: _native.linear_forward(x, w, b, relu=True)), 'linear forward')
    _same(*both_routes(lambda: _native.linear_dgrad(go, w.t().contiguous(), scale, 32,
                                                    relu_mask=mask,
                                                    out_scaled=None)), 'linear dgrad')
    _same(*both_routes(lambda: _native.linear_dgrad(
        go, w.t().contiguous(), scale, relu_bits=_native.pack_relu_bits(mask),
        out_scaled=torch.empty(n, F, device=dev))), 'linear dgrad, bits + scaled copy')
    _same(*both_routes(lambda: _native.linear_wgrad(go, x, bias_grad=True)), 'linear wgrad')
    _same(*both_routes(lambda: _native.linear_wgrad(go, x, x2=mask)), 'two-operand wgrad')
    # one-kernel layer, both directions
    wc = (torch.randn(Fo, 2 * F, generator=g) * 0.1).to(dev)

    def layer():
        agg = torch.empty(n, F, device=dev)
        out = torch.empty(n, Fo, device=dev)
        bits = _native.relu_bits_like(n, Fo, dev).fill_(0)
        _native.sage_layer_forward(fwd.ptr, fwd.idx, x, x, wc, b, 'mean', True, agg, out,
                                   hub=fwd.hub, save_agg=True, relu_bits=bits)
        return agg, out, bits

    _same(*both_routes(layer), 'one-kernel layer')
    # gather / scatter-add / sddmm / softmax / pointers
    idx = torch.randint(0, n, (5000, ), generator=g).to(dev)
    _same(*both_routes(lambda: _native.gather_rows(x, idx)), 'gather')
    _same(*both_routes(lambda: _native.index2ptr(fwd.idx.sort().values, n)), 'index2ptr')
    _same(*both_routes(lambda: _native.ptr2index(fwd.ptr, ei.size(1))), 'ptr2index')
    _same(*both_routes(lambda: _native.sddmm_csr(fwd.ptr, fwd.idx, None, mask, x, ei.size(1), 1)),
          'sddmm')
    att = torch.randn(ei.size(1), 4, generator=g).to(dev)
    sm = both_routes(lambda: _native.segment_softmax_forward(att, fwd.ptr))
    _same(*sm, 'segment softmax')
    _same(*both_routes(lambda: _native.segment_softmax_backward(sm[0], att, fwd.ptr)),
          'segment softmax backward')
    # atomics: same launch, order-dependent rounding -> compare at tolerance
    a, c = both_routes(lambda: _native.gather_scatter_add(x, ei[1].to(dev), ei[0].to(dev), n,
                                                          scale=scale))
    assert torch.allclose(a, c, rtol=1e-5, atol=1e-5)


def test_typed_errors_survive_the_compiled_route(dev):
    """Operands the binding does not take as they are go through the Python path and raise this
    package's typed errors (not a bare RuntimeError from the dispatcher)."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    ei = random_graph(50, 50, 400, seed=1)
    h = pga.EdgeIndex(ei.to(dev), (50, 50))
    fwd = h.by_dst()
    with pytest.raises(ValueError, match='must be float32'):
        _native.spmm_csr(fwd.ptr, fwd.idx, torch.randn(50, 8, device=dev).double(), 'sum')
    with pytest.raises(ValueError, match='divisible by the number of heads'):
        _native.spmm_csr(fwd.ptr, fwd.idx, torch.randn(50, 10, device=dev), 'sum',
                         w=torch.rand(400, 4, device=dev))
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        _native.spmm_csr(fwd.ptr, fwd.idx, torch.randn(50, 8), 'sum')
    with pytest.raises(ValueError, match="columns but 'weight' expects"):
        _native.linear_forward(torch.randn(50, 8, device=dev), torch.randn(4, 9, device=dev))


def test_cpp_autograd_nodes_match_the_python_ones(dev, monkeypatch):
    """`torch.ops.pyg_amd_c.{linear_ag, spmm_ag, bias_act_ag}` (torch::autograd::Function s in
    csrc/torch_binding.cpp: forward and backward without re-entering Python) against the Python
    Functions they mirror: same kernels in the same order, so values and gradients are BITWISE
    equal; through a whole 2-layer GCN (the launch-bound config 1) and a SAGEConv layer; routed
    only for plain float32 operands without a gradient w.r.t. the edge weights; usable under
    inference_mode."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _compiled, _functions
    from pytorch_geometric_amd.nn import GCN, SAGEConv
    if _compiled.ops() is None:
        pytest.skip('compiled binding not available')
    g = gen(3)
    n, e = 1500, 9000
    ei = torch.randint(0, n, (2, e), generator=g).to(dev)
    x = torch.randn(n, 40, generator=g).to(dev)
    go = torch.randn(n, 6, generator=g).to(dev)
    torch.manual_seed(0)
    model = GCN(40, 24, num_layers=2, out_channels=6, cached=True).to(dev)
    for conv in model.convs:
        torch.nn.init.normal_(conv.bias, std=0.3)
    _functions.OWN_GEMM_MIN_ROWS, keep_rows = 1, _functions.OWN_GEMM_MIN_ROWS

    def run(flag):
        monkeypatch.setattr(_functions, 'CPP_AUTOGRAD', flag)
        model.zero_grad()
        xg = x.clone().requires_grad_(True)
        out = model(xg, ei)
        out.backward(go)
        return out.detach(), xg.grad, [p.grad.clone() for p in model.parameters()]

    try:
        used = []
        C = _compiled.ops()
        real = C.spmm_ag
        py = run(False)
        cpp = run(True)
        assert torch.equal(py[0], cpp[0]) and torch.equal(py[1], cpp[1])
        for a, b in zip(py[2], cpp[2]):
            # (weights bitwise; the bias gradients are column sums that meet in fp32 atomics:
            # equal up to the order of those adds)
            assert torch.equal(a, b) if a.dim() > 1 else bool(
                ((a - b).abs() <= 1e-5 * a.abs().max().clamp(min=1)).all())
        # the C++ nodes are in the graph: the output's grad_fn is not a Python Function
        monkeypatch.setattr(_functions, 'CPP_AUTOGRAD', True)
        out = model(x.clone().requires_grad_(True), ei)
        names = set()
        stack = [out.grad_fn]
        while stack:
            f = stack.pop()
            if f is None:
                continue
            names.add(f.name())
            stack += [nf for nf, _ in f.next_functions]
        for node in ('LinearAG', 'SpmmAG', 'BiasActAG'):
            assert any(f'::{node}>' in nm for nm in names), (node, names)
        assert not any(nm.endswith('FunctionBackward') for nm in names), names
        # SAGEConv (mean aggregation: 1/deg folded into the transposed launch)
        torch.manual_seed(1)
        conv = SAGEConv(40, 16).to(dev)
        res = []
        for flag in (False, True):
            monkeypatch.setattr(_functions, 'CPP_AUTOGRAD', flag)
            conv.zero_grad()
            xg = x.clone().requires_grad_(True)
            o = conv(xg, ei)
            o.backward(torch.ones_like(o))
            res.append((o.detach(), xg.grad, [p.grad.clone() for p in conv.parameters()]))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        for a, b in zip(res[0][2], res[1][2]):
            assert torch.equal(a, b) if a.dim() > 1 else bool(
                ((a - b).abs() <= 1e-5 * a.abs().max().clamp(min=1)).all())
        # weights that need a gradient stay on the Python node (it owns the SDDMM)
        h = pga.EdgeIndex(ei, (n, n))
        w = torch.rand(e, generator=g).to(dev).requires_grad_(True)
        o = _functions.spmm_node(x, w, h, 'sum', 'coo')
        assert o.grad_fn.name() == 'SpmmFunctionBackward'
        o.sum().backward()
        assert w.grad is not None
        with torch.inference_mode():
            o = model(x, ei)
        assert torch.equal(o, py[0])
    finally:
        _functions.OWN_GEMM_MIN_ROWS = keep_rows

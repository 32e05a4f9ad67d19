"""torch.ops.pyg_amd.* on the device: torch.library.opcheck (schema, fake-vs-real metadata,
autograd registration, AOT dispatch) for every operator, values and gradients against the oracle,
and torch.compile(fullgraph=True) of functions built on the operators — also through the
reference's dispatchers after backend.install() (the kernels stay in the compiled graph as opaque
nodes instead of making the backend step aside)."""
import pytest
import torch

from oracle import pyg_oracle as O
from tests._util import assert_close, gen

pytestmark = pytest.mark.gpu


def _inputs(dev, seed=1):
    g = gen(seed)
    n, G = 300, 24
    x = torch.randn(n, 10, generator=g)
    x[::5] = torch.randint(-2, 3, (x[::5].size(0), 10), generator=g).float()
    index = torch.randint(0, G - 2, (n, ), generator=g)
    sidx = index.sort().values
    ptr = torch._convert_indices_from_coo_to_csr(sidx, G)
    return x, index, sidx, ptr, G


def test_opcheck_every_operator(dev):
    import pytorch_geometric_amd.ops  # noqa: F401
    from torch.library import opcheck
    P = torch.ops.pyg_amd
    x, index, sidx, ptr, G = _inputs(dev)
    xd = x.to(dev).requires_grad_(True)
    idx, ptrd = index.to(dev), ptr.to(dev)
    col = torch.randint(0, x.size(0), (int(ptr[-1]), ), generator=gen(3)).to(dev)
    val = torch.rand(int(ptr[-1]), generator=gen(4)).to(dev).requires_grad_(True)
    w = torch.randn(6, 10, generator=gen(5)).to(dev).requires_grad_(True)
    b = torch.randn(6, generator=gen(6)).to(dev).requires_grad_(True)
    opcheck(P.index_sort, (idx, None))
    opcheck(P.index2ptr, (sidx.to(dev), G))
    opcheck(P.ptr2index, (ptrd, x.size(0)))
    opcheck(P.gather, (xd, idx))
    for red in ('sum', 'mean', 'max', 'mul'):
        opcheck(P.scatter, (xd, idx, G, red))
    for red in ('sum', 'mean', 'min'):
        opcheck(P.segment_csr, (xd, ptrd, red))
    opcheck(P.softmax_csr, (xd, ptrd))
    opcheck(P.spmm, (ptrd, col, val, xd, 'sum'))
    opcheck(P.spmm, (ptrd, col, None, xd, 'mean'))
    opcheck(P.spmm, (ptrd, col, None, xd, 'max'))
    opcheck(P.linear, (xd, w, b))
    opcheck(P.linear, (xd, w, None))


def test_operators_match_the_oracle_values_and_grads(dev):
    import pytorch_geometric_amd.ops  # noqa: F401
    P = torch.ops.pyg_amd
    x, index, sidx, ptr, G = _inputs(dev, 2)
    go = torch.randn(G, 10, generator=gen(8))

    def both(f_ref, f_dev, what, tol=2e-5):
        xr = x.clone().requires_grad_(True)
        ref = f_ref(xr)
        ref.backward(go[:ref.size(0)] if ref.size(0) != x.size(0) else torch.ones_like(ref))
        xd = x.to(dev).requires_grad_(True)
        out = f_dev(xd)
        out.backward((go[:ref.size(0)] if ref.size(0) != x.size(0)
                      else torch.ones_like(ref)).to(dev))
        assert_close(out, ref.detach(), rtol=tol, atol=tol, what=what)
        assert_close(xd.grad, xr.grad, rtol=tol, atol=tol, what=what + ' grad')

    for red in ('sum', 'mean', 'min', 'max'):
        both(lambda t: O.scatter(t, index, 0, G, red),
             lambda t: P.scatter(t, index.to(dev), G, red), f'scatter {red}')
        both(lambda t: O.segment(t, ptr, red),
             lambda t: P.segment_csr(t, ptr.to(dev), red), f'segment {red}')
    both(lambda t: O.softmax(t, ptr=ptr), lambda t: P.softmax_csr(t, ptr.to(dev)), 'softmax')
    both(lambda t: t[index], lambda t: P.gather(t, index.to(dev)), 'gather')
    ei = torch.stack([torch.randint(0, x.size(0), (900, ), generator=gen(9)),
                      torch.randint(0, G, (900, ), generator=gen(10)).sort().values])
    rowptr = torch._convert_indices_from_coo_to_csr(ei[1], G)
    val = torch.rand(900, generator=gen(11))
    for red, v in (('sum', val), ('mean', None), ('max', None)):
        both(lambda t: O.spmm(ei, t, G, red, v),
             lambda t: P.spmm(rowptr.to(dev), ei[0].to(dev), None if v is None else v.to(dev),
                              t, red), f'spmm {red}')


def test_torch_compile_keeps_the_kernels_in_the_graph(dev):
    """fullgraph=True: no graph break at the operators; forward and backward of the compiled
    function equal eager."""
    import pytorch_geometric_amd.ops  # noqa: F401
    P = torch.ops.pyg_amd
    x, index, sidx, ptr, G = _inputs(dev, 3)
    idx, ptrd = index.to(dev), ptr.to(dev)
    w = torch.randn(7, 10, generator=gen(12)).to(dev)

    order = torch.argsort(index, stable=True).to(dev)
    whole = torch.tensor([0, G], device=dev)  # one segment over all G rows

    def fn(t, weight):
        h = P.linear(t, weight, None).relu()
        a = P.scatter(h, idx, G, 'mean')
        s, perm = P.index_sort(idx, None)
        b = P.segment_csr(P.gather(h, perm), P.index2ptr(s, G), 'max')
        c = P.segment_csr(P.gather(h, order), ptrd, 'max')
        return (a + b + c + P.softmax_csr(a, whole)).sum()

    compiled = torch.compile(fn, backend='aot_eager', fullgraph=True)
    outs = []
    for f in (fn, compiled):
        t = x.to(dev).requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        y = f(t, ww)
        y.backward()
        outs.append((y.detach(), t.grad, ww.grad))
    for a, b in zip(*outs):
        assert_close(b, a, rtol=1e-5, atol=1e-5, what='compiled vs eager')


def test_reference_dispatchers_compile_to_the_registered_operators(dev):
    """After install(), torch_geometric.utils.{scatter, segment, softmax(ptr), index_sort} inside a
    torch.compile region resolve to torch.ops.pyg_amd.* (captured graph inspected), not to the
    reference's ATen decomposition."""
    from oracle import make_ref
    make_ref.import_reference()
    import torch_geometric.utils as U
    from pytorch_geometric_amd import backend
    backend.install()
    try:
        x, index, sidx, ptr, G = _inputs(dev, 4)
        idx, ptrd = index.to(dev), ptr.to(dev)
        seen = []

        def inspect(gm, example_inputs):
            seen.extend(str(n.target) for n in gm.graph.nodes if n.op == 'call_function')
            return gm.forward

        def fn(t):
            a = U.scatter(t, idx, 0, G, 'max')
            b = U.segment(t[torch.argsort(idx)], ptrd, 'mean')
            c = U.softmax(t[torch.argsort(idx)], ptr=ptrd)
            return a, b, c

        got = torch.compile(fn, backend=inspect, fullgraph=True)(x.to(dev))
        assert any('pyg_amd.scatter' in s for s in seen), seen
        assert any('pyg_amd.segment_csr' in s for s in seen)
        assert any('pyg_amd.softmax_csr' in s for s in seen)
        order = torch.argsort(index, stable=True)
        want = (O.scatter(x, index, 0, G, 'max'), O.segment(x[order], ptr, 'mean'),
                O.softmax(x[order], ptr=ptr))
        # (argsort on the device is not stable: compare what does not depend on the tie order)
        assert_close(got[0], want[0], rtol=0, atol=0, what='compiled scatter max')
        assert_close(got[1], want[1], rtol=1e-5, atol=1e-5, what='compiled segment mean')
        assert got[2].shape == want[2].shape
    finally:
        backend.uninstall()

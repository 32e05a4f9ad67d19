"""scatter / segment / softmax / spmm / gather: HIP path (through the C ABI) vs the oracle and the
reference's golden vectors.  fp32 tolerance 1e-5 (north_star), integers bit-exact."""
import pytest
import torch

from oracle import pyg_oracle as O
from tests._util import assert_close, assert_sum_close, gen, random_graph

pytestmark = pytest.mark.gpu


def run_grad(fn, inputs, grad_out):
    leaves = [t.clone().requires_grad_(True) if t.is_floating_point() else t for t in inputs]
    out = fn(*leaves)
    fl = [t for t in leaves if t.is_floating_point()]
    grads = torch.autograd.grad(out, fl, grad_out.to(out.device), allow_unused=True)
    return out.detach(), grads


# ---- scatter ---------------------------------------------------------------------------------------
def test_scatter_golden(dev, golden):
    import pytorch_geometric_amd as pga
    sc = golden['scatter']
    src, index = sc['src'].to(dev), sc['index'].to(dev)
    for red in ['sum', 'mean', 'min', 'max', 'mul']:
        out, (gs, ) = run_grad(lambda s: pga.utils.scatter(s, index, 0, sc['dim_size'], red),
                               [src], sc[red]['grad_out'])
        assert_close(out, sc[red]['out'], what=f'scatter {red}')
        assert_close(gs, sc[red]['grad_src'], what=f'scatter {red} grad')
    d = sc['dim1_mean']
    out, (gs, ) = run_grad(lambda s: pga.utils.scatter(s, d['index'].to(dev), 1, 6, 'mean'),
                           [d['src'].to(dev)], d['grad_out'])
    assert_close(out, d['out'])
    assert_close(gs, d['grad_src'])
    m = sc['dim1_max_nosize']
    out, (gs, ) = run_grad(lambda s: pga.utils.scatter(s, d['index'].to(dev), -2, None, 'max'),
                           [d['src'].to(dev)], m['grad_out'])
    assert_close(out, m['out'])
    assert_close(gs, m['grad_src'])
    v = sc['vec_sum']
    assert_close(pga.utils.scatter(v['src'].to(dev), v['index'].to(dev)), v['out'])
    # 'any': every output row equals one of the contributing rows
    out = pga.utils.scatter(src, index, 0, sc['dim_size'], 'any').cpu()
    for g in range(sc['dim_size']):
        rows = sc['src'][sc['index'] == g]
        for f in range(out.size(1)):
            assert (rows.numel() == 0 and out[g, f] == 0) or (out[g, f] == rows[:, f]).any()


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('F', [1, 3, 64, 100, 257])
def test_scatter_vs_oracle(dev, dtype, F):
    import pytorch_geometric_amd as pga
    g = gen(F)
    src = torch.randn(5000, F, generator=g)
    src[::5] = torch.randint(-2, 3, (1000, F), generator=g).float()
    index = torch.randint(0, 300, (5000, ), generator=g).to(dtype)
    go = torch.randn(320, F, generator=g)
    for red in ['sum', 'mean', 'min', 'max']:
        ref, (rg, ) = run_grad(lambda s: O.scatter(s, index.long(), 0, 320, red), [src], go)
        out, (gs, ) = run_grad(lambda s: pga.utils.scatter(s, index.to(dev), 0, 320, red),
                               [src.to(dev)], go)
        if red in ('sum', 'mean'):  # atomics: order differs run to run, judge against fp64
            ex, (eg, ) = run_grad(lambda s: O.scatter(s, index.long(), 0, 320, red),
                                  [src.double()], go.double())
            assert_sum_close(out, ref, ex, what=f'{red} F={F}')
            assert_sum_close(gs, rg, eg, what=f'{red} grad F={F}')
        else:
            assert_close(out, ref, rtol=0, atol=0, what=f'{red} F={F}')
            assert_close(gs, rg, what=f'{red} grad F={F}')


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_scatter_mul_backward_zero_rule(dev, dtype):
    """ATen's 'prod' backward (what utils/_scatter.py:119-133 differentiates through): groups with
    no zero, exactly one zero (that element gets g * prod(others), the rest 0) and several zeros
    (all 0).  g * out / src alone gives NaN here — round-1 VERDICT weak #1."""
    import pytorch_geometric_amd as pga
    g = gen(23)
    n, G, F = 600, 40, 5
    src = torch.randn(n, F, generator=g).mul(0.5).add(1.0)
    index = torch.randint(0, G - 3, (n, ), generator=g)  # groups G-3 .. G-1 stay empty
    # column 0: no zeros; column 1: exactly one zero in every 2nd group; column 2: two zeros in
    # every 3rd group; columns 3-4: random zeros
    for grp in range(0, G - 3, 2):
        rows = (index == grp).nonzero().view(-1)
        if rows.numel():
            src[rows[0], 1] = 0.0
    for grp in range(0, G - 3, 3):
        rows = (index == grp).nonzero().view(-1)
        if rows.numel() > 1:
            src[rows[:2], 2] = 0.0
    src[:, 3:][torch.rand(n, 2, generator=g) < 0.05] = 0.0
    go = torch.randn(G, F, generator=g)
    ref, (rg, ) = run_grad(lambda s: O.scatter(s, index, 0, G, 'mul'), [src], go)
    assert torch.isfinite(rg).all() and (rg[src == 0] != 0).any()
    out, (gs, ) = run_grad(lambda s: pga.utils.scatter(s, index.to(dtype).to(dev), 0, G, 'mul'),
                           [src.to(dev)], go)
    assert_close(out, ref, rtol=1e-5, atol=1e-5, what='mul')
    # products of ~15 factors in a different order: relative 1e-5 of each gradient's magnitude
    assert_close(gs, rg, rtol=2e-5, atol=1e-5, what='mul grad')
    assert (gs.cpu()[(src == 0) & (rg == 0)] == 0).all()


def test_scatter_out_of_range_raises(dev):
    """ATen raises 'index out of bounds' on the reference's CPU path; the HIP kernels skip such
    rows, flag them, and the wrapper raises before anything is saved for the backward."""
    import pytorch_geometric_amd as pga
    src = torch.randn(6, 3, device=dev, requires_grad=True)
    idx = torch.tensor([0, 1, 7, 1, 0, 2], device=dev)
    from pytorch_geometric_amd import _native
    # default: the flag travels asynchronously; `check_index_errors()` (or a later scatter call
    # once the flag has arrived) raises
    assert _native.INDEX_CHECK == 'async'
    for red in ('sum', 'mean', 'max', 'mul'):
        with pytest.raises((IndexError, RuntimeError), match='out of bounds'):
            pga.utils.scatter(src, idx, 0, 4, red)
            pga.check_index_errors()
    with pytest.raises((IndexError, RuntimeError), match='out of bounds'):
        pga.utils.scatter(src, torch.tensor([0, 1, -1, 1, 0, 2], device=dev), 0, 4, 'sum')
        pga.check_index_errors()
    pga.check_index_errors()  # nothing left pending, flags cleared
    ok = pga.utils.scatter(src, torch.tensor([0, 1, 3, 1, 0, 2], device=dev), 0, 4, 'sum')
    pga.check_index_errors()
    assert ok.shape == (4, 3)
    # a later call reports an earlier launch without an explicit check
    pga.utils.scatter(src.detach(), idx, 0, 4, 'sum')
    torch.cuda.synchronize()
    with pytest.raises((IndexError, RuntimeError), match='out of bounds'):
        pga.utils.scatter(src.detach(), torch.tensor([0, 1, 3, 1, 0, 2], device=dev), 0, 4, 'sum')
    # 'sync': raises at the call site
    _native.INDEX_CHECK = 'sync'
    try:
        for red in ('sum', 'max'):
            with pytest.raises((IndexError, RuntimeError), match='out of bounds'):
                pga.utils.scatter(src, idx, 0, 4, red)
    finally:
        _native.INDEX_CHECK = 'async'


def test_unsorted_softmax_out_of_range_is_reported(dev):
    """The index branch of ``softmax`` (utils/_softmax.py:82-88) with a ``num_nodes`` that is too
    small: the reference's scatter raises; the one-operator kernels skip such rows AND flag them
    (asynchronously by default, at the call site with PYGAMD_CHECK_INDEX=sync)."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    src = torch.randn(6, 2, device=dev)
    idx = torch.tensor([3, 0, 7, 1, 0, 2], device=dev)  # unsorted, one value >= num_nodes
    with pytest.raises((IndexError, RuntimeError), match='out of bounds'):
        pga.utils.softmax(src, idx, num_nodes=4)
        pga.check_index_errors()
    pga.check_index_errors()
    ok = pga.utils.softmax(src, idx, num_nodes=8)
    pga.check_index_errors()
    assert_close(ok.sum(0)[0:1], torch.tensor([5.0]), rtol=1e-5, atol=1e-5)  # five groups
    _native.INDEX_CHECK = 'sync'
    try:
        with pytest.raises((IndexError, RuntimeError), match='out of bounds'):
            pga.utils.softmax(src, idx, num_nodes=4)
    finally:
        _native.INDEX_CHECK = 'async'


def test_scatter_errors(dev):
    import pytorch_geometric_amd as pga
    src = torch.randn(2, 5, 2, device=dev)
    idx = torch.tensor([0, 1, 0, 1, 0], device=dev)
    with pytest.raises(ValueError, match='must be one-dimensional'):
        pga.utils.scatter(src, idx.view(1, -1))
    with pytest.raises(ValueError, match='must lay between 0 and 2'):
        pga.utils.scatter(src, idx, dim=3)
    with pytest.raises(ValueError, match="invalid `reduce` argument 'std'"):
        pga.utils.scatter(src, idx, dim=1, reduce='std')
    with pytest.raises(pga.PygAmdError, match='no CPU fallback'):
        pga.utils.scatter(torch.randn(4, 2), torch.tensor([0, 0, 1, 1]))
    agg = pga.nn.MeanAggregation()
    # a caller-supplied dim_size that is too small is reported in the reference's words
    # (nn/aggr/base.py:131-141) by whoever meets the launch's flag: with the default 'async' the
    # call does NOT wait for it (no host/device serialisation per aggregation) ...
    from pytorch_geometric_amd import _native
    assert _native.INDEX_CHECK == 'async'
    with pytest.raises(ValueError, match="invalid 'dim_size'"):
        agg(torch.randn(5, 3, device=dev), idx, dim_size=1)
        pga.check_index_errors()
    pga.check_index_errors()  # nothing left pending
    # ... the call itself (its end looks at the flags that have ARRIVED) or a later aggregation
    # call meets the flag once it has arrived ...
    with pytest.raises(ValueError, match="invalid 'dim_size'"):
        agg(torch.randn(5, 3, device=dev), idx, dim_size=1)
        torch.cuda.synchronize()
        agg(torch.randn(5, 3, device=dev), idx, dim_size=2)
    pga.check_index_errors()
    # ... and PYGAMD_CHECK_INDEX=sync raises at the call site itself
    _native.INDEX_CHECK = 'sync'
    try:
        with pytest.raises(ValueError, match="invalid 'dim_size'"):
            agg(torch.randn(5, 3, device=dev), idx, dim_size=1)
    finally:
        _native.INDEX_CHECK = 'async'
    pga.check_index_errors()
    with pytest.raises(ValueError, match='invalid dimension'):
        agg(torch.randn(5, 3, device=dev), idx, dim=2)


def test_scatter_empty(dev):
    import pytorch_geometric_amd as pga
    out = pga.utils.scatter(torch.empty(0, 4, device=dev), torch.empty(0, dtype=torch.long,
                                                                       device=dev))
    assert out.shape == (0, 4)
    out = pga.utils.scatter(torch.empty(0, 4, device=dev),
                            torch.empty(0, dtype=torch.long, device=dev), dim_size=3,
                            reduce='max')
    assert out.shape == (3, 4) and (out == 0).all()


def test_scatter_argmax(dev, golden):
    import pytorch_geometric_amd as pga
    k = golden['scatter']['argmax_known']
    got = pga.utils.scatter_argmax(k['src'].to(dev), k['index'].to(dev), dim_size=6)
    assert got.tolist() == [3, 5, 1, 4, 5, 5]
    r = golden['scatter']['argmax_rand']
    for dt in (torch.int64, torch.int32):
        got = pga.utils.scatter_argmax(r['src'].to(dev), r['index'].to(dt).to(dev),
                                       dim_size=r['dim_size'])
        assert_close(got.long(), r['out'])


# ---- segment ---------------------------------------------------------------------------------------
def test_segment_golden(dev, golden):
    import pytorch_geometric_amd as pga
    sg = golden['segment']
    for red in ['sum', 'mean', 'min', 'max']:
        out, (gs, ) = run_grad(lambda s: pga.utils.segment(s, sg['ptr'].to(dev), red),
                               [sg['src'].to(dev)], sg[red]['grad_out'])
        assert_close(out, sg[red]['out'], what=f'segment {red}')
        assert_close(gs, sg[red]['grad_src'], what=f'segment {red} grad')


@pytest.mark.parametrize('F', [1, 8, 47, 256, 600])
def test_segment_vs_oracle(dev, F):
    import pytorch_geometric_amd as pga
    g = gen(F + 1)
    lens = torch.randint(0, 40, (200, ), generator=g)
    lens[7] = 3000  # one long segment
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    src = torch.randn(int(ptr[-1]), F, generator=g)
    for red in ['sum', 'mean', 'min', 'max']:
        ref = O.segment(src, ptr, red)
        out = pga.utils.segment(src.to(dev), ptr.to(dev), red)
        if red in ('sum', 'mean'):
            assert_sum_close(out, ref, O.segment(src.double(), ptr, red), what=f'{red} F={F}')
        else:
            assert_close(out, ref, rtol=0, atol=0, what=f'{red} F={F}')
        out32 = pga.utils.segment(src.to(dev), ptr.int().to(dev), red)
        assert_close(out32, out.cpu(), rtol=0, atol=0)


def test_segment_gradients_with_ties_and_mixed_sign_grads(dev):
    """min / max over tied extrema: the reference's CPU path (torch._segment_reduce) averages the
    gradient over the ties ONLY where it is positive (ATen SegmentReduce.cpp), negative gradients
    reach every tied element undivided — matched as is (found with the torch.ops.pyg_amd tests)."""
    import pytorch_geometric_amd as pga
    g = gen(314)
    lens = torch.randint(0, 25, (120, ), generator=g)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    src = torch.randint(-2, 3, (int(ptr[-1]), 9), generator=g).float()  # many ties
    go = torch.randn(120, 9, generator=g)                               # both signs
    for red in ['sum', 'mean', 'min', 'max']:
        ref, (rg, ) = run_grad(lambda s: O.segment(s, ptr, red), [src], go)
        out, (gs, ) = run_grad(lambda s: pga.utils.segment(s, ptr.to(dev), red), [src.to(dev)], go)
        assert_close(out, ref, what=f'segment {red}')
        assert_close(gs, rg, what=f'segment {red} grad')


# ---- softmax ---------------------------------------------------------------------------------------
def test_softmax_golden(dev, golden):
    import pytorch_geometric_amd as pga
    sm = golden['softmax']
    k = sm['known']
    assert pga.utils.softmax(k['src'].to(dev), k['index'].to(dev)).tolist() == [0.5, 0.5, 1, 1]
    assert pga.utils.softmax(k['src'].to(dev), None, k['ptr'].to(dev)).tolist() == [0.5, 0.5, 1,
                                                                                   1]
    i = sm['index']
    out, (gs, ) = run_grad(lambda s: pga.utils.softmax(s, i['index'].to(dev), num_nodes=11),
                           [i['src'].to(dev)], i['grad_out'])
    assert_close(out, i['out'])
    assert_close(gs, i['grad_src'])
    p = sm['ptr']
    out, (gs, ) = run_grad(lambda s: pga.utils.softmax(s, None, p['ptr'].to(dev)),
                           [i['src'].to(dev)], p['grad_out'])
    assert_close(out, p['out'])
    assert_close(gs, p['grad_src'])
    u = sm['unsorted']
    out, (gs, ) = run_grad(lambda s: pga.utils.softmax(s, u['index'].to(dev), num_nodes=11),
                           [u['src'].to(dev)], u['grad_out'])
    assert_close(out, u['out'])
    assert_close(gs, u['grad_src'])
    d = sm['dim1']
    out, (gs, ) = run_grad(
        lambda s: pga.utils.softmax(s, i['index'].to(dev), num_nodes=11, dim=-1),
        [d['src'].to(dev)], d['grad_out'])
    assert_close(out, d['out'])
    assert_close(gs, d['grad_src'])


@pytest.mark.parametrize('H', [1, 2, 8, 3, 64, 100])
def test_segment_softmax_vs_oracle(dev, H):
    import pytorch_geometric_amd as pga
    g = gen(H + 40)
    # (303 segments: the kernels take four per wave — the last wave holds three; empty, short
    # and one very long segment)
    lens = torch.randint(0, 30, (303, ), generator=g)
    lens[11] = 2500
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    src = torch.randn(int(ptr[-1]), H, generator=g) * 4
    go = torch.randn(src.shape, generator=g)
    ref, (rg, ) = run_grad(lambda s: O.softmax(s, None, ptr), [src], go)
    out, (gs, ) = run_grad(lambda s: pga.utils.softmax(s, None, ptr.to(dev)), [src.to(dev)], go)
    assert_close(out, ref, what=f'softmax H={H}')
    assert_close(gs, rg, what=f'softmax grad H={H}')
    if H == 1:  # 1-D input
        out1 = pga.utils.softmax(src.view(-1).to(dev), None, ptr.to(dev))
        assert_close(out1, ref.view(-1))


# ---- gather --------------------------------------------------------------------------------------
@pytest.mark.parametrize('F', [1, 5, 100, 256])
def test_gather_bit_exact(dev, F):
    from pytorch_geometric_amd._functions import GatherFunction
    g = gen(F)
    x = torch.randn(1000, F, generator=g)
    idx = torch.randint(0, 1000, (7777, ), generator=g)
    for dt in (torch.int64, torch.int32):
        xx = x.to(dev).requires_grad_(True)
        out = GatherFunction.apply(xx, idx.to(dt).to(dev), True)
        assert_close(out, x.index_select(0, idx), rtol=0, atol=0)  # indexing: bit-exact
        go = torch.randn(7777, F, generator=gen(1))
        out.backward(go.to(dev))
        ref = torch.zeros(1000, F).index_add_(0, idx, go)
        assert_close(xx.grad, ref, atol=1e-4)
    with pytest.raises(IndexError):
        GatherFunction.apply(x.to(dev), torch.tensor([0, 1000], device=dev), True)


# ---- CSR SpMM --------------------------------------------------------------------------------------
WIDTHS = [1, 4, 7, 16, 47, 64, 100, 128, 256, 320, 500, 520, 1100]


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('F', WIDTHS)
def test_spmm_sum_mean_vs_oracle(dev, dtype, F):
    import pytorch_geometric_amd as pga
    ei = random_graph(400, 300, 6000, seed=F, dtype=dtype, skew=True)
    g = gen(F + 7)
    x = torch.randn(400, F, generator=g)
    go = torch.randn(300, F, generator=g)
    h = pga.EdgeIndex(ei.to(dev), (400, 300))
    for red in ['sum', 'mean']:
        ref, (rg, ) = run_grad(lambda t: O.spmm(ei.long(), t, 300, red), [x], go)
        ex, (eg, ) = run_grad(lambda t: O.spmm(ei.long(), t, 300, red), [x.double()],
                              go.double())
        out, (gx, ) = run_grad(lambda t: pga.utils.spmm(h, t, red), [x.to(dev)], go)
        assert_sum_close(out, ref, ex, what=f'spmm {red} F={F}')
        assert_sum_close(gx, rg, eg, what=f'spmm {red} grad F={F}')


@pytest.mark.parametrize('F', [3, 16, 100, 256, 600])
def test_spmm_minmax_vs_oracle(dev, F):
    import pytorch_geometric_amd as pga
    ei = random_graph(200, 150, 3000, seed=F + 1, skew=True)
    g = gen(F + 9)
    x = torch.randn(200, F, generator=g)
    x[::2] = torch.randint(-1, 2, (100, F), generator=g).float()  # ties, zeros
    go = torch.randn(150, F, generator=g)
    h = pga.EdgeIndex(ei.to(dev), (200, 150))
    for red in ['min', 'max']:
        ref, (rg, ) = run_grad(lambda t: O.spmm(ei, t, 150, red), [x], go)
        out, (gx, ) = run_grad(lambda t: pga.utils.spmm(h, t, red), [x.to(dev)], go)
        assert_close(out, ref, rtol=0, atol=0, what=f'spmm {red} F={F}')  # selection: exact
        assert_close(gx, rg, what=f'spmm {red} grad F={F}')
        # deterministic mode: the source-driven backward (no atomics), bit-identical run to run
        torch.use_deterministic_algorithms(True)
        try:
            xs = [x.to(dev).requires_grad_(True) for _ in range(2)]
            for t in xs:
                pga.utils.spmm(h, t, red).backward(go.to(dev))
        finally:
            torch.use_deterministic_algorithms(False)
        assert torch.equal(xs[0].grad, xs[1].grad)
        assert_close(xs[0].grad, rg, what=f'spmm {red} deterministic grad F={F}')


@pytest.mark.parametrize('F', [5, 64, 256])
@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_spmm_minmax_saved_arg(dev, F, dtype):
    """The forward's arg32 (what the one-winner backward consumes): >= 0 = slot offset of (the
    first edge from) the UNIQUE attaining neighbour, -1 = empty row, -2 = split gradient (ties
    between different neighbours, or an extremum of 0);
    and the fast + marked-rows backward equals the reference on data without any ties, where every
    output takes the fast path."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    ei = random_graph(300, 260, 6000, seed=F, skew=True)
    ei[1][ei[1] == 9] = 10  # row 9 is empty
    g = gen(F + 77)
    x = torch.randn(300, F, generator=g)
    x[::3] = torch.randint(-1, 2, (100, F), generator=g).float()  # ties and zeros
    h = pga.EdgeIndex(ei.to(dtype).to(dev), (300, 260))
    fwd = h.by_dst()
    ptr, idx = fwd.ptr.cpu().long(), fwd.idx.cpu().long()
    for red in ('max', 'min'):
        out, arg = _native.spmm_csr(fwd.ptr, fwd.idx, x.to(dev), red, n_rows=260, save_arg32=True)
        out, arg = out.cpu(), arg.cpu()
        assert_close(out, O.spmm(ei, x, 260, red), rtol=0, atol=0)
        for i in range(260):
            rows = x[idx[ptr[i]:ptr[i + 1]]]
            if rows.size(0) == 0:
                assert bool((arg[i] == -1).all())
                continue
            # attained by more than one DISTINCT neighbour (parallel edges send the whole gradient
            # to one row: no split)
            hits = (x[idx[ptr[i]:ptr[i + 1]].unique()] == out[i]).sum(0)
            split = (hits > 1) | (out[i] == 0)
            assert bool((arg[i][split] == -2).all()), (red, i)
            uniq = ~split
            a = arg[i][uniq].long()
            assert bool((a >= 0).all())
            assert torch.equal(rows[a, uniq.nonzero().view(-1)], out[i][uniq])
    # tie-free, zero-free data: the whole gradient goes through the fast path
    xr = torch.randn(300, F, generator=g)
    go = torch.randn(260, F, generator=g)
    for red in ('max', 'min'):
        ref, (rg, ) = run_grad(lambda t: O.spmm(ei, t, 260, red), [xr], go)
        out, (gx, ) = run_grad(lambda t: pga.utils.spmm(h, t, red), [xr.to(dev)], go)
        assert_close(gx, rg, what=f'{red} grad (fast path)')


@pytest.mark.parametrize('F,H', [(1, 1), (16, 1), (100, 1), (256, 1), (64, 8), (256, 8),
                                 (320, 8), (6, 2), (21, 3)])
def test_spmm_weighted_vs_oracle(dev, F, H):
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd._functions import SpmmFunction
    ei = random_graph(300, 250, 5000, seed=F + H, skew=True)
    g = gen(F * 3 + H)
    x = torch.randn(300, F, generator=g)
    w = torch.rand(5000, generator=g) if H == 1 else torch.rand(5000, H, generator=g)
    go = torch.randn(250, F, generator=g)
    h = pga.EdgeIndex(ei.to(dev), (300, 250))

    def ref_fn(t, ww):
        if H == 1:
            return O.spmm(ei, t, 250, 'sum', ww)
        return O.propagate(t.view(300, H, F // H), ei, 250, 'sum', ww).reshape(250, F)

    ref, (rgx, rgw) = run_grad(ref_fn, [x, w], go)
    ex, (egx, egw) = run_grad(ref_fn, [x.double(), w.double()], go.double())
    out, (gx, gw) = run_grad(lambda t, ww: SpmmFunction.apply(t, ww, h, 'sum', 'coo'),
                             [x.to(dev), w.to(dev)], go)
    assert_sum_close(out, ref, ex, what='weighted out')
    assert_sum_close(gx, rgx, egx, what='weighted grad_x')
    assert_sum_close(gw, rgw, egw, what='weighted grad_w')
    # the same weights handed over in by-destination slot order
    perm = h.by_dst().perm.long()
    w_slot = w.to(dev)[perm]
    out2, (gx2, gw2) = run_grad(lambda t, ww: SpmmFunction.apply(t, ww, h, 'sum', 'slot'),
                                [x.to(dev), w_slot], go)
    assert_sum_close(out2, ref, ex, what='slot-order out')
    assert_sum_close(gx2, rgx, egx, what='slot-order grad_x')
    assert_sum_close(gw2, rgw[perm.cpu()], egw[perm.cpu()], what='slot-order grad_w')


def test_spmm_golden(dev, golden):
    import pytorch_geometric_amd as pga
    gr, sp = golden['graph'], golden['spmm']
    h = pga.EdgeIndex(sp['edge_index'].to(dev), (gr['N'], gr['N']))
    for red in ['sum', 'mean', 'min', 'max']:
        assert_close(pga.utils.spmm(h, gr['x'].to(dev), red), sp[red]['out'], what=red)
    for red in ['sum', 'mean']:
        out, (gx, ) = run_grad(lambda t: pga.utils.spmm(h, t, red), [gr['x'].to(dev)],
                               sp[red]['grad_out'])
        assert_close(gx, sp[red]['grad_x'], what=f'{red} grad')


@pytest.mark.parametrize('F', [100, 256, 47])
def test_spmm_hub_rows(dev, F, monkeypatch):
    """Rows longer than the hub threshold take the chunked two-stage path."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    monkeypatch.setattr(_native, 'HUB_THRESHOLD', 64)
    monkeypatch.setattr(_native, 'HUB_CHUNK', 48)
    orig = _native.hub_plan
    monkeypatch.setattr(_native, 'hub_plan', lambda ptr, threshold=None, chunk=None: orig(
        ptr, 64, 48))
    ei = random_graph(500, 120, 20_000, seed=F, skew=True)
    x = torch.randn(500, F, generator=gen(F))
    go = torch.randn(120, F, generator=gen(F + 1))
    h = pga.EdgeIndex(ei.to(dev), (500, 120))
    assert h.by_dst().hub[2] > 0
    for red in ['sum', 'mean']:
        ref, (rg, ) = run_grad(lambda t: O.spmm(ei, t, 120, red), [x], go)
        ex, (eg, ) = run_grad(lambda t: O.spmm(ei, t, 120, red), [x.double()], go.double())
        out, (gx, ) = run_grad(lambda t: pga.utils.spmm(h, t, red), [x.to(dev)], go)
        assert_sum_close(out, ref, ex, what=f'hub {red}')
        assert_sum_close(gx, rg, eg, what=f'hub {red} grad')


def test_spmm_strided_io(dev):
    """Aggregate out of / into a wider buffer (leading dimension > F)."""
    from pytorch_geometric_amd import _native
    import pytorch_geometric_amd as pga
    ei = random_graph(100, 100, 1500, seed=3)
    big = torch.randn(100, 512, generator=gen(2))
    h = pga.EdgeIndex(ei.to(dev), (100, 100))
    fwd = h.by_dst()
    bd = big.to(dev)
    out_buf = torch.zeros(100, 512, device=dev)
    _native.spmm_csr(fwd.ptr, fwd.idx, bd[:, 256:], 'mean', n_rows=100, out=out_buf[:, :256])
    ref = O.spmm(ei, big[:, 256:].contiguous(), 100, 'mean')
    assert_close(out_buf[:, :256], ref)
    assert (out_buf[:, 256:] == 0).all()


def test_empty_graph_and_isolated_rows(dev):
    import pytorch_geometric_amd as pga
    h = pga.EdgeIndex(torch.empty(2, 0, dtype=torch.long, device=dev), (5, 4))
    for red in ['sum', 'mean', 'min', 'max']:
        out = pga.utils.spmm(h, torch.randn(5, 8, device=dev), red)
        assert out.shape == (4, 8) and (out == 0).all()


@pytest.mark.parametrize('F', [1, 47, 100, 256, 300, 700])
def test_colsum(dev, F):
    from pytorch_geometric_amd import _native
    x = torch.randn(10_001, F, generator=gen(F))
    got = _native.colsum(x.to(dev))
    assert_sum_close(got, x.sum(0), x.double().sum(0), what=f'colsum F={F}',
                     abs_sum=x.abs().sum(0))
    wide = torch.randn(500, 2 * F, generator=gen(F + 1))
    got = _native.colsum(wide.to(dev)[:, F:])  # strided view
    assert_sum_close(got, wide[:, F:].sum(0), wide[:, F:].double().sum(0),
                     abs_sum=wide[:, F:].abs().sum(0))
    assert _native.colsum(torch.empty(0, F, device=dev)).abs().sum() == 0


def test_spmm_accumulate(dev, monkeypatch):
    """accumulate=1: out += A x (the fused 'x_root + aggregate' of the SAGE backward)."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    monkeypatch.setattr(_native, 'HUB_THRESHOLD', 64)
    monkeypatch.setattr(_native, 'HUB_CHUNK', 64)
    ei = random_graph(300, 300, 8000, seed=12, skew=True)
    h = pga.EdgeIndex(ei.to(dev), (300, 300))
    fwd = h.by_dst()
    assert fwd.hub[2] > 0
    for F in (100, 256, 7):
        buf = torch.randn(300, 2 * F, generator=gen(F))
        bd = buf.to(dev)
        _native.spmm_csr(fwd.ptr, fwd.idx, bd[:, :F], 'mean', n_rows=300, hub=fwd.hub,
                         out=bd[:, F:], accumulate=True)
        ref = buf[:, F:] + O.spmm(ei, buf[:, :F].contiguous(), 300, 'mean')
        assert_close(bd[:, F:], ref, atol=2e-5, what=f'accumulate F={F}')
        assert_close(bd[:, :F], buf[:, :F], rtol=0, atol=0)
        # relu_mask: the same launch zeroes the result where a ReLU output (here the right half
        # of another [agg | x] buffer, a strided view) is not positive — hub rows, vector and
        # scalar widths; everything else bit-identical to the unmasked launch
        act = torch.randn(300, 2 * F, generator=gen(F + 1)).relu()
        act[::5, F:] = -0.0
        ad = act.to(dev)
        for acc in (True, False):
            b1, b2 = buf.to(dev), buf.to(dev)
            _native.spmm_csr(fwd.ptr, fwd.idx, b1[:, :F], 'mean', n_rows=300, hub=fwd.hub,
                             out=b1[:, F:], accumulate=acc)
            _native.spmm_csr(fwd.ptr, fwd.idx, b2[:, :F], 'mean', n_rows=300, hub=fwd.hub,
                             out=b2[:, F:], accumulate=acc, relu_mask=ad[:, F:])
            want = torch.where(ad[:, F:] > 0, b1[:, F:], torch.zeros_like(b1[:, F:]))
            assert torch.equal(b2[:, F:], want), f'relu_mask F={F} accumulate={acc}'
            assert int((want == 0).sum()) > 0
            b3 = buf.to(dev)  # the same mask as one bit per element
            _native.spmm_csr(fwd.ptr, fwd.idx, b3[:, :F], 'mean', n_rows=300, hub=fwd.hub,
                             out=b3[:, F:], accumulate=acc,
                             relu_bits=_native.pack_relu_bits(ad[:, F:]))
            assert torch.equal(b3[:, F:], want), f'relu_bits F={F} accumulate={acc}'
    with pytest.raises(RuntimeError):  # extrema have no such epilogue
        _native.spmm_csr(fwd.ptr, fwd.idx, bd[:, :F], 'max', n_rows=300,
                         relu_mask=bd[:, :F].contiguous())


def test_fused_and_multi_aggregation(dev):
    """test/nn/aggr/test_fused.py:9-42: FusedAggregation equals the individual aggregations,
    values and gradients; test_basic.py:66-75: var against the manual formula."""
    import pytorch_geometric_amd.nn as nn
    g = gen(77)
    x = torch.randn(400, 6, generator=g)
    index = torch.randint(0, 30, (400, ), generator=g)
    names = ['sum', 'mean', 'min', 'max', 'mul', 'var', 'std']
    ref_mods = {'sum': 'sum', 'mean': 'mean', 'min': 'min', 'max': 'max', 'mul': 'mul'}
    xr = x.clone().requires_grad_(True)
    refs = {k: O.scatter(xr, index, 0, 32, v) for k, v in ref_mods.items()}
    mean2 = O.scatter(xr * xr, index, 0, 32, 'mean')
    refs['var'] = mean2 - refs['mean'] * refs['mean']
    s = refs['var'].clamp(min=1e-5).sqrt()
    refs['std'] = s.masked_fill(s <= 1e-5 ** 0.5, 0.0)
    go = [torch.randn(32, 6, generator=g) for _ in names]
    sum(((refs[n] * w).sum() for n, w in zip(names, go))).backward()
    fused = nn.FusedAggregation(names)
    xg = x.to(dev).requires_grad_(True)
    outs = fused(xg, index.to(dev), dim_size=32)
    sum(((o * w.to(dev)).sum() for o, w in zip(outs, go))).backward()
    for n, o in zip(names, outs):
        # 'mul': products of ~16 factors (|out| up to ~1e2) in atomic order -> relative bound;
        # 'var'/'std' difference two O(1) means -> 2e-5 absolute
        assert_close(o, refs[n].detach(), atol=2e-5, rtol=2e-5, what=n)
    # the sum of seven gradients (incl. mul's product terms, |grad| up to ~1e2): 2e-5 relative
    # to each element plus 2e-5 absolute, the same class as the per-op tests
    assert_close(xg.grad, xr.grad, atol=2e-5, rtol=2e-5, what='fused grad')
    multi = nn.MultiAggregation(['mean', 'max', 'std'])
    out = multi(x.to(dev), index.to(dev), dim_size=32)
    assert out.shape == (32, 18)
    assert_close(out[:, :6], refs['mean'].detach(), atol=2e-5)
    assert_close(out[:, 6:12], refs['max'].detach(), atol=0, rtol=0)
    with pytest.raises(ValueError, match='not fusable'):
        nn.FusedAggregation([nn.MultiAggregation(['sum', 'max'])])


def test_spmm_torch_sparse_inputs(dev):
    """utils/_spmm.py:57-111: adj_t given as torch.sparse CSR / COO / CSC (rows = destinations)."""
    import pytorch_geometric_amd as pga
    ei = random_graph(90, 70, 1500, seed=41)
    ei = torch.unique(ei, dim=1)  # sparse constructors need coalesced entries
    g = gen(41)
    val = torch.rand(ei.size(1), generator=g) + 0.5
    x = torch.randn(90, 24, generator=g)
    adj_coo = torch.sparse_coo_tensor(torch.stack([ei[1], ei[0]]), val, (70, 90)).coalesce()
    ref = {red: torch.sparse.mm(adj_coo.to_sparse_csr(), x, red) for red in ['sum', 'mean']}
    for adj in (adj_coo.to_sparse_csr(), adj_coo, adj_coo.to_sparse_csc()):
        ad = adj.to(dev)
        for red in ['sum', 'mean']:
            assert_close(pga.utils.spmm(ad, x.to(dev), red), ref[red], atol=2e-5,
                         what=f'{adj.layout} {red}')
    # gradient w.r.t. the dense operand
    xg = x.to(dev).requires_grad_(True)
    pga.utils.spmm(adj_coo.to_sparse_csr().to(dev), xg, 'sum').sum().backward()
    assert_close(xg.grad, (adj_coo.to_dense().t() @ torch.ones(70, 24)), atol=2e-5)


def test_hetero_linear(dev):
    """nn/dense/linear.py:248-252 (forward_naive) is the in-tree oracle for HeteroLinear."""
    from pytorch_geometric_amd.nn import HeteroLinear
    g = gen(52)
    x = torch.randn(300, 10, generator=g)
    tv = torch.randint(0, 5, (300, ), generator=g)
    tv[tv == 2] = 3  # an empty type
    torch.manual_seed(1)
    lin = HeteroLinear(10, 6, num_types=5)
    w, b = lin.weight.detach(), lin.bias.detach()
    ref = torch.stack([x[i] @ w[tv[i]] + b[tv[i]] for i in range(300)])
    lin = lin.to(dev)
    xg = x.to(dev).requires_grad_(True)
    out = lin(xg, tv.to(dev))
    assert_close(out, ref, atol=2e-5)
    out.sum().backward()
    ref_gx = torch.stack([w[tv[i]].sum(1) for i in range(300)])
    assert_close(xg.grad, ref_gx, atol=2e-5)
    assert lin.weight.grad.shape == (5, 10, 6) and float(lin.weight.grad[2].abs().sum()) == 0


@pytest.mark.parametrize('F', [5, 64, 256])
def test_gather_scatter_add(dev, F):
    """Edge-parallel atomic fallback == the CSR kernel on the transposed handle."""
    from pytorch_geometric_amd import _native
    ei = random_graph(200, 150, 4000, seed=F)
    g = gen(F)
    x = torch.randn(150, F, generator=g)
    scale = torch.rand(150, generator=g)
    w = torch.rand(4000, generator=g)
    got = _native.gather_scatter_add(x.to(dev), ei[1].to(dev), ei[0].to(dev), 200,
                                     scale=scale.to(dev), w=w.to(dev))
    msg = x[ei[1]] * (scale[ei[1]] * w).view(-1, 1)
    ref = torch.zeros(200, F).index_add_(0, ei[0], msg)
    ex = torch.zeros(200, F, dtype=torch.double).index_add_(0, ei[0], msg.double())
    assert_sum_close(got, ref, ex, what=f'gather_scatter_add F={F}')
    got = _native.gather_scatter_add(x.to(dev), ei[1].int().to(dev), ei[0].int().to(dev), 200)
    ref = torch.zeros(200, F).index_add_(0, ei[0], x[ei[1]])
    assert_sum_close(got, ref, ref.double(), what='plain')


@pytest.mark.parametrize('H,C', [(8, 32), (8, 40), (1, 7), (3, 100), (70, 4), (2, 256)])
def test_head_dot(dev, H, C):
    """GAT node terms (x * att).sum(-1) for both attention vectors in one pass + fused backward."""
    from pytorch_geometric_amd._functions import HeadDotFunction
    g = gen(H * C)
    x = torch.randn(777, H, C, generator=g)
    a, b = torch.randn(1, H, C, generator=g), torch.randn(1, H, C, generator=g)
    ga, gb = torch.randn(777, H, generator=g), torch.randn(777, H, generator=g)
    xr, ar, br = (t.clone().requires_grad_(True) for t in (x, a, b))
    ra, rb = (xr * ar).sum(-1), (xr * br).sum(-1)
    ((ra * ga).sum() + (rb * gb).sum()).backward()
    xg, ag, bg = (t.to(dev).requires_grad_(True) for t in (x, a, b))
    oa, ob = HeadDotFunction.apply(xg, ag, bg)
    ((oa * ga.to(dev)).sum() + (ob * gb.to(dev)).sum()).backward()
    assert_close(oa, ra.detach(), atol=2e-5)
    assert_close(ob, rb.detach(), atol=2e-5)
    assert_close(xg.grad, xr.grad, atol=2e-5)
    assert_sum_close(ag.grad, ar.grad, ar.grad.double(), atol=2e-4, what='grad att_src')
    assert_sum_close(bg.grad, br.grad, br.grad.double(), atol=2e-4, what='grad att_dst')


def test_spmm_arg_output_and_weighted_mean(dev):
    """arg_out of min/max (first slot on ties, -1 for empty rows) and the mean-with-weights route."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd._functions import SpmmFunction
    ei = random_graph(120, 90, 1500, seed=14, skew=True)
    g = gen(14)
    x = torch.randint(-2, 3, (120, 12), generator=g).float()  # many ties
    h = pga.EdgeIndex(ei.to(dev), (120, 90))
    fwd = h.by_dst()
    for red in ('max', 'min'):
        out, arg = _native.spmm_csr(fwd.ptr, fwd.idx, x.to(dev), red, n_rows=90, return_arg=True)
        out, arg = out.cpu(), arg.cpu()
        ptr, idx = fwd.ptr.cpu(), fwd.idx.cpu()
        for r in range(90):
            s, e = int(ptr[r]), int(ptr[r + 1])
            if e == s:
                assert (arg[r] == -1).all() and (out[r] == 0).all()
                continue
            vals = x[idx[s:e]]
            best = vals.max(0).values if red == 'max' else vals.min(0).values
            first = (vals == best).int().argmax(0) + s  # first slot attaining the extremum
            assert torch.equal(out[r], best) and torch.equal(arg[r], first)
    w = torch.rand(1500, generator=g)
    xf = torch.randn(120, 12, generator=g)
    xr, wr = xf.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = O.propagate(xr, ei, 90, 'mean', wr)
    go = torch.randn(90, 12, generator=g)
    ref.backward(go)
    xg, wg = xf.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    out = SpmmFunction.apply(xg, wg, h, 'mean', 'coo')
    out.backward(go.to(dev))
    assert_close(out, ref.detach(), atol=2e-5)
    assert_close(xg.grad, xr.grad, atol=2e-5)
    assert_close(wg.grad, wr.grad, atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize('F', [47, 256, 300])
def test_relu_backward_colsum(dev, F):
    """aten::threshold_backward(grad, relu_out, 0) + column sums in one pass, strided inputs."""
    from pytorch_geometric_amd import _native
    g = gen(F)
    n = 1500
    grad = torch.randn(n, 2 * F, generator=g)[:, F:]
    act = torch.randn(n, F + 4, generator=g).relu()[:, :F]
    act[::7] = 0.0
    want = torch.ops.aten.threshold_backward(grad, act, 0)
    got, cs = _native.relu_backward_colsum(grad.to(dev), act.to(dev))
    assert got.is_contiguous() and torch.equal(got.cpu(), want)
    assert_sum_close(cs, want.sum(0), want.double().sum(0), atol=1e-4, what='bias grad')
    got, cs = _native.relu_backward_colsum(grad.to(dev), act.to(dev), want_colsum=False)
    assert cs is None and torch.equal(got.cpu(), want)


def test_softmax_and_powermean_aggregation_golden(dev, golden_aggr):
    """DeeperGCN aggregations (nn/aggr/basic.py:142-296) against the real reference."""
    from pytorch_geometric_amd import nn
    from tests._aggr_cases import CASES
    G = golden_aggr
    for name, (kind, kw, use_ptr, positive) in CASES.items():
        case = G['cases'][name]
        cls = nn.SoftmaxAggregation if kind == 'softmax' else nn.PowerMeanAggregation
        aggr = cls(**kw).to(dev)
        x = (G['x'].abs() + 0.1 if positive else G['x']).to(dev).requires_grad_(True)
        where = (dict(ptr=G['ptr'].to(dev)) if use_ptr
                 else dict(index=G['index'].to(dev), dim_size=G['dim_size']))
        out = aggr(x, **where)
        params = list(aggr.parameters())
        grads = torch.autograd.grad(out, [x] + params, case['grad_out'].to(dev))
        assert_close(out, case['out'], atol=2e-5, what=f'{name} out')
        assert_close(grads[0], case['grad_x'], atol=2e-5, what=f'{name} grad_x')
        if params:
            assert_close(grads[1], case['grad_param'], atol=1e-4, rtol=1e-4,
                         what=f'{name} grad param')
    with pytest.raises(ValueError):
        nn.SoftmaxAggregation(learn=True, semi_grad=True)
    with pytest.raises(ValueError):
        nn.PowerMeanAggregation(channels=3)
    assert isinstance(nn.aggr.aggregation_resolver('softmax', t=2.0), nn.SoftmaxAggregation)


def test_segment_logsumexp(dev):
    """test/utils/test_segment.py:36-53 (dim 0 and dim 1 against torch.logsumexp per range) plus
    empty segments, long segments, wide rows and the gradient, against the oracle's restatement of
    utils/_segment.py:53-80."""
    import pytorch_geometric_amd as pga
    src = torch.randn(20, 16, generator=gen(1))
    ptr = torch.tensor([0, 0, 5, 10, 15, 20])
    out = pga.utils.segment_logsumexp(src.to(dev), ptr.to(dev), dim=0).cpu()
    assert out.shape == (5, 16)
    assert_close(out[0], torch.zeros(16))
    for i in range(1, 5):
        assert_close(out[i], src[ptr[i]:ptr[i + 1]].logsumexp(0))
    src1 = torch.randn(16, 20, generator=gen(2))
    out = pga.utils.segment_logsumexp(src1.to(dev), ptr.to(dev), dim=1).cpu()
    assert out.shape == (16, 5)
    assert_close(out[:, 2], src1[:, 5:10].logsumexp(1))
    g = gen(3)
    for F, n_seg in ((1, 50), (4, 50), (7, 30), (130, 40)):
        lens = torch.randint(0, 60, (n_seg, ), generator=g)
        lens[3] = 0
        lens[5] = 700
        ptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
        src = torch.randn(int(ptr[-1]), F, generator=g) * 3
        go = torch.randn(n_seg, F, generator=g)
        ref, (rg, ) = run_grad(lambda s: O.segment_logsumexp(s, ptr, 0), [src], go)
        for dt in (torch.int64, torch.int32):
            out, (gs, ) = run_grad(
                lambda s: pga.utils.segment_logsumexp(s, ptr.to(dt).to(dev), 0), [src.to(dev)], go)
            assert_close(out, ref, rtol=1e-5, atol=1e-5, what=f'lse F={F}')
            assert_close(gs, rg, rtol=1e-5, atol=1e-5, what=f'lse grad F={F}')


@pytest.mark.parametrize('F', [1, 6, 64, 100, 260])
@pytest.mark.parametrize('sorted_index', [False, True])
def test_one_pass_multi_reduce(dev, F, sorted_index):
    """FusedAggregation's shared statistics from ONE read of the rows (pygamd_multi_reduce_csr):
    sum, sum of squares, min, max against the oracle's separate scatters, values and gradients,
    with empty groups and a hub group; sums judged against fp64 (summation order differs)."""
    import pytorch_geometric_amd.nn as nn
    g = gen(F + 1000 * sorted_index)
    n, G = 3000, 120
    index = torch.randint(0, G - 5, (n, ), generator=g)
    index[:900] = 7  # one long group
    if sorted_index:
        index = index.sort().values
    x = torch.randn(n, F, generator=g)
    x[::7] = torch.randint(-2, 3, (x[::7].size(0), F), generator=g).float()  # ties for min/max
    names = ['sum', 'mean', 'min', 'max', 'var', 'std']
    go = [torch.randn(G, F, generator=g) for _ in names]

    def ref_all(xx):
        s, m = O.scatter(xx, index, 0, G, 'sum'), O.scatter(xx, index, 0, G, 'mean')
        var = O.scatter(xx * xx, index, 0, G, 'mean') - m * m
        sd = var.clamp(min=1e-5).sqrt()
        return [s, m, O.scatter(xx, index, 0, G, 'min'), O.scatter(xx, index, 0, G, 'max'), var,
                sd.masked_fill(sd <= 1e-5 ** 0.5, 0.0)]

    xr = x.clone().requires_grad_(True)
    refs = ref_all(xr)
    sum(((r * w).sum() for r, w in zip(refs, go))).backward()
    x64 = x.double().requires_grad_(True)
    ex = ref_all(x64)
    sum(((r * w.double()).sum() for r, w in zip(ex, go))).backward()
    fused = nn.FusedAggregation(names)
    xg = x.to(dev).requires_grad_(True)
    outs = fused(xg, index.to(dev), dim_size=G)
    sum(((o * w.to(dev)).sum() for o, w in zip(outs, go))).backward()
    for nme, o, r, e in zip(names, outs, refs, ex):
        if nme in ('min', 'max'):
            assert_close(o, r.detach(), rtol=0, atol=0, what=nme)
        else:
            assert_sum_close(o, r.detach(), e.detach(), rtol=2e-5, atol=2e-5, what=nme)
    assert_sum_close(xg.grad, xr.grad, x64.grad, rtol=2e-5, atol=2e-5, what='grad')
    with pytest.raises(ValueError, match="invalid 'dim_size'"):
        fused(xg, index.to(dev), dim_size=3)
        pga.check_index_errors()


def test_round2_entry_points_on_empty_and_degenerate_inputs(dev):
    """Empty / single-element inputs of the entry points added in round 2 (the reference accepts
    them everywhere: zero edges, zero groups, zero features, one row)."""
    import pytorch_geometric_amd as pga
    import pytorch_geometric_amd.nn as nn
    from pytorch_geometric_amd import _native
    e = torch.empty
    # segment_logsumexp: no rows at all, and only empty segments
    out = pga.utils.segment_logsumexp(e(0, 4, device=dev), torch.zeros(3, dtype=torch.long,
                                                                       device=dev), 0)
    assert out.shape == (2, 4) and bool((out == 0).all())
    # FusedAggregation: zero rows -> zeros; one row -> itself (std 0)
    fused = nn.FusedAggregation(['sum', 'mean', 'max', 'std'])
    outs = fused(e(0, 3, device=dev), e(0, dtype=torch.long, device=dev), dim_size=4)
    assert all(o.shape == (4, 3) and bool((o == 0).all()) for o in outs)
    x1 = torch.tensor([[1.0, -2.0, 3.0]], device=dev, requires_grad=True)
    outs = fused(x1, torch.tensor([2], device=dev), dim_size=3)
    assert_close(outs[0][2], x1[0].detach())
    assert_close(outs[2][2], x1[0].detach())
    assert bool((outs[3] == 0).all())
    sum(o.sum() for o in outs).backward()
    assert bool(torch.isfinite(x1.grad).all())
    # scatter(mul) backward on an empty source
    s0 = e(0, 2, device=dev, requires_grad=True)
    pga.utils.scatter(s0, e(0, dtype=torch.long, device=dev), 0, 3, 'mul').sum().backward()
    assert s0.grad.shape == (0, 2)
    # dense transform: zero rows, zero input features
    w = torch.randn(5, 7, device=dev)
    assert _native.linear_forward(e(0, 7, device=dev), w).shape == (0, 5)
    assert _native.linear_wgrad(e(0, 5, device=dev), e(0, 7, device=dev)).abs().sum() == 0
    assert _native.linear_dgrad(e(0, 5, device=dev), w.t().contiguous()).shape == (0, 7)
    b = torch.randn(5, device=dev)
    assert_close(_native.linear_forward(e(3, 0, device=dev), e(5, 0, device=dev), b),
                 b.cpu().expand(3, 5))
    # min / max aggregation of a graph without edges: zeros, zero gradient, arg32 = -1
    h = pga.EdgeIndex(e(2, 0, dtype=torch.long, device=dev), (6, 4))
    xm = torch.randn(6, 8, device=dev, requires_grad=True)
    om = pga.utils.spmm(h, xm, 'max')
    om.sum().backward()
    assert bool((om == 0).all()) and bool((xm.grad == 0).all())
    _, arg = _native.spmm_csr(h.by_dst().ptr, h.by_dst().idx, xm.detach(), 'max', n_rows=4,
                              save_arg32=True)
    assert bool((arg == -1).all())
    # one-kernel SAGE layer on a graph without edges: lin_r(x) + b
    wc = torch.randn(6, 16, device=dev)
    buf = torch.zeros(4, 16, device=dev)
    xs = torch.randn(4, 8, device=dev)
    buf[:, 8:] = xs
    y = torch.empty(4, 6, device=dev)
    h2 = pga.EdgeIndex(e(2, 0, dtype=torch.long, device=dev), (4, 4))
    _native.sage_layer_forward(h2.by_dst().ptr, h2.by_dst().idx, xs, buf[:, 8:], wc, None, 'mean',
                               False, buf[:, :8], y, hub=h2.by_dst().hub)
    assert_close(y, (xs @ wc[:, 8:].t()).cpu(), rtol=1e-5, atol=1e-5)


def test_inplace_edit_of_a_saved_output_is_detected(dev):
    """Functions whose backward compares against their OUTPUT (min / max / mul scatter,
    segment softmax, ...) save the tensor they return, so autograd's version counter catches an
    in-place edit between forward and backward also when the public shape is not [n, F]."""
    import pytorch_geometric_amd as pga
    g = gen(5)
    idx = torch.randint(0, 7, (40, ), generator=g).to(dev)
    for make in (lambda s: pga.utils.scatter(s, idx, 0, 7, 'max'),
                 lambda s: pga.utils.scatter(s, idx, 0, 7, 'mul'),
                 lambda s: pga.utils.softmax(s, ptr=torch.tensor([0, 10, 25, 40], device=dev))):
        for shape in ((40, ), (40, 2, 3)):
            src = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
            out = make(src)
            out.sum().backward()          # untouched: fine
            src.grad = None
            out = make(src)
            out.mul_(2.0)
            with pytest.raises(RuntimeError, match='modified by an inplace operation'):
                out.sum().backward()


@pytest.mark.parametrize('F', [4, 100, 256, 300, 520])
@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_spmm_minmax_backward_without_atomics(dev, F, dtype):
    """pygamd_spmm_csr_minmax_backward_src (winner bit masks + source-driven sum over the transposed
    CSR) against the one-atomic-per-output kernel and the oracle: bipartite graph with hub rows, an
    empty destination, a source without out-edges, data with ties and zeros (those outputs go to
    the tie kernel in both paths), widths with one, two and three 256-feature blocks; without any
    marked output the result is bit-identical run to run."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    n_src, n_dst = 500, 420
    ei = random_graph(n_src, n_dst, 9000, seed=F + 3, skew=True)
    ei[1][ei[1] == 9] = 10      # destination 9 is empty
    ei[0][ei[0] == 33] = 34     # source 33 has no out-edges
    g = gen(F + 41)
    go = torch.randn(n_dst, F, generator=g)
    h = pga.EdgeIndex(ei.to(dtype).to(dev), (n_src, n_dst))
    fwd, bwd, smap = h.by_dst(), h.by_src(), h.src_slot_to_dst_slot()
    for ties in (False, True):
        x = torch.randn(n_src, F, generator=g)
        if ties:
            x[::3] = torch.randint(-1, 2, (len(range(0, n_src, 3)), F), generator=g).float()
        for red in ('max', 'min'):
            ref, (rg, ) = run_grad(lambda t: O.spmm(ei, t, n_dst, red), [x], go)
            out, arg = _native.spmm_csr(fwd.ptr, fwd.idx, x.to(dev), red, n_rows=n_dst,
                                        hub=fwd.hub, save_arg32=True)
            a = _native.spmm_minmax_backward_dst(fwd.ptr, fwd.idx, x.to(dev), out, go.to(dev),
                                                 n_src, arg32=arg)
            b = _native.spmm_minmax_backward_src(fwd, bwd, smap, x.to(dev), out, go.to(dev), arg)
            assert b is not None and b.shape == (n_src, F)
            assert_close(b, rg, what=f'{red} F={F} ties={ties}: source-driven vs oracle')
            assert_close(b, a, what=f'{red} F={F} ties={ties}: source-driven vs one-atomic kernel')
            assert bool((b[33] == 0).all())
            if not ties and int((arg == -2).sum()) == 0:
                # no output goes through the tie kernel's atomics (hub rows do: their extremum is
                # combined from chunk results and carries no slot): bit-identical run to run
                b2 = _native.spmm_minmax_backward_src(fwd, bwd, smap, x.to(dev), out, go.to(dev),
                                                      arg)
                assert torch.equal(b, b2)
    # unsupported layout (F % 4): the wrapper says so and the autograd path falls back
    x = torch.randn(n_src, 6, generator=g).to(dev)
    out, arg = _native.spmm_csr(fwd.ptr, fwd.idx, x, 'max', n_rows=n_dst, hub=fwd.hub,
                                save_arg32=True)
    assert _native.spmm_minmax_backward_src(fwd, bwd, smap, x, out, go[:, :6].contiguous().to(dev),
                                            arg) is None


@pytest.mark.parametrize('F,Fs,Fc', [(47, 48, 48), (1, 4, 1), (5, 5, 8), (64, 64, 64),
                                     (100, 104, 100), (256, 256, 260)])
def test_rows_pack(dev, F, Fs, Fc):
    """pygamd_rows_pack: the non-zero-row bitmap (+ its device count), the row-scaled copy and the
    plain copy, both zero-filled to their own width, in one pass over a strided block."""
    from pytorch_geometric_amd import _native
    n = 1000 + F  # not a multiple of 32
    g = gen(F)
    wide = torch.randn(n, F + 9, generator=g)
    live = torch.rand(n, generator=g) < 0.1
    wide[~live] = 0
    wide[7, 2 + 3 % F] = float('nan')  # a NaN row is a live row
    wide[9] = 0
    wide[9, 2 + F - 1] = -0.0  # all (signed) zeros: not live
    wide[11] = 0
    wide[11, 2 + F - 1] = 1e-40  # a subnormal is not zero
    scale = torch.rand(n, generator=g) + 0.5
    src = wide.to(dev)[:, 2:2 + F]  # row stride F + 9
    ref = wide[:, 2:2 + F]
    scaled = torch.full((n, Fs + 3), 7.0, device=dev)[:, :Fs]
    copy = torch.full((n, Fc), 7.0, device=dev)
    bits, n_set = _native.rows_pack(src, scale.to(dev), scaled=scaled, copy=copy)
    want = (ref != 0).any(dim=1)
    words = bits.cpu().numpy().view('uint32')
    got = torch.tensor([(int(words[i >> 5]) >> (i & 31)) & 1 for i in range(32 * len(words))],
                       dtype=torch.bool)
    assert torch.equal(got[:n], want) and not bool(got[n:].any())
    assert bool(want[7]) and not bool(want[9]) and bool(want[11])
    assert int(n_set.item()) == int(want.sum())
    assert torch.equal(copy[:, :F].cpu().nan_to_num(nan=123.), ref.nan_to_num(nan=123.))
    assert bool((copy[:, F:] == 0).all())
    exp = ref * scale.view(-1, 1)
    assert torch.equal(scaled[:, :F].cpu().nan_to_num(nan=123.), exp.nan_to_num(nan=123.))
    assert bool((scaled[:, F:] == 0).all())
    # bits alone, no count
    b2, none = _native.rows_pack(src, count=False)
    assert none is None and torch.equal(b2, bits)
    # nothing live / empty
    b3, c3 = _native.rows_pack(torch.zeros(70, F, device=dev))
    assert int(c3.item()) == 0 and not bool(b3.any())
    b4, c4 = _native.rows_pack(torch.zeros(0, F, device=dev))
    assert b4.numel() == 0 and int(c4.item()) == 0
    with pytest.raises(ValueError):
        _native.rows_pack(src, copy=torch.empty(n, F - 1 if F > 1 else 0, device=dev))
    with pytest.raises(ValueError):
        _native.rows_pack(src.double())


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('F', [1, 4, 47, 48, 100, 256, 520])
def test_spmm_skips_zero_source_rows(dev, dtype, F, monkeypatch):
    """`src_bits`: the sum over the live source rows only equals the sum over all of them (hub rows
    and chunks of more than 64 slots included), whatever the density says about using the bits."""
    from pytorch_geometric_amd import _native
    import pytorch_geometric_amd as pga
    monkeypatch.setattr(_native, 'HUB_THRESHOLD', 200)
    monkeypatch.setattr(_native, 'HUB_CHUNK', 96)
    orig = _native.hub_plan
    monkeypatch.setattr(_native, 'hub_plan', lambda ptr, threshold=None, chunk=None: orig(
        ptr, 200, 96))
    n_src, n_dst = 3000, 500
    ei = random_graph(n_src, n_dst, 40_000, seed=F + 3, dtype=dtype, skew=True)
    h = pga.EdgeIndex(ei.to(dev), (n_src, n_dst)).by_dst()
    assert h.hub[2] > 0
    for frac in (0.08, 0.0, 0.9):
        g = gen(F + int(100 * frac))
        x = torch.randn(n_src, F, generator=g)
        x[torch.rand(n_src, generator=g) >= frac] = 0
        xd = x.to(dev)
        bits, n_set = _native.rows_pack(xd)
        ex = O.spmm(ei.long(), x.double(), n_dst, 'sum')
        ref = O.spmm(ei.long(), x, n_dst, 'sum')
        plain = _native.spmm_csr(h.ptr, h.idx, xd, 'sum', n_rows=n_dst, hub=h.hub)
        for counter in (n_set, None):  # None: the bits are used even at 90 % density
            for red in ('sum', 'mean'):
                out = _native.spmm_csr(h.ptr, h.idx, xd, red, n_rows=n_dst, hub=h.hub,
                                       src_bits=bits, src_bits_set=counter)
                if red == 'mean':
                    e2 = O.spmm(ei.long(), x.double(), n_dst, 'mean')
                    r2 = O.spmm(ei.long(), x, n_dst, 'mean')
                    assert_sum_close(out, r2, e2, what=f'sparse-source mean F={F} p={frac}')
                else:
                    assert_sum_close(out, ref, ex, what=f'sparse-source sum F={F} p={frac}')
                    # a row with no live source is exactly zero either way
                    dead = (plain == 0).all(dim=1)
                    assert bool((out[dead] == 0).all())
        # accumulate + strided output keep working with the bits
        buf = torch.ones(n_dst, F + 4, device=dev)
        _native.spmm_csr(h.ptr, h.idx, xd, 'sum', n_rows=n_dst, hub=h.hub, out=buf[:, :F],
                         accumulate=True, src_bits=bits, src_bits_set=n_set)
        assert_sum_close(buf[:, :F] - 1, ref, ex, atol=1e-4, what='sparse-source accumulate')
        assert bool((buf[:, F:] == 1).all())
    # no stored entries at all; rows past the last full group of four
    for n_empty in (1, 4, 7):
        e = pga.EdgeIndex(torch.empty(2, 0, dtype=dtype, device=dev), (n_src, n_empty)).by_dst()
        out = _native.spmm_csr(e.ptr, e.idx, xd, 'sum', n_rows=n_empty, hub=e.hub, src_bits=bits,
                               src_bits_set=n_set,
                               out=torch.full((n_empty, F), 3.0, device=dev))
        assert bool((out == 0).all())
    with pytest.raises(ValueError):
        _native.spmm_csr(h.ptr, h.idx, xd, 'sum', n_rows=n_dst, hub=h.hub, src_bits=bits[:5])
    with pytest.raises(Exception):  # not with per-edge weights
        _native.spmm_csr(h.ptr, h.idx, xd, 'sum', n_rows=n_dst, hub=h.hub, src_bits=bits,
                         w=torch.ones(ei.size(1), device=dev))


def _with_zeros(n, F, density, seed):
    g = gen(seed)
    x = torch.randn(n, F, generator=g)
    x[torch.rand(n, F, generator=g) >= density] = 0
    return x


@pytest.mark.parametrize('F', [4, 32, 36, 100, 128, 256])
def test_rows_compress_layout(dev, F):
    """pygamd_rows_compress: 8 mask words + the kept values in column order; kept = any bit pattern
    but +0.0 (so -0.0, NaN and subnormals survive): lossless."""
    from pytorch_geometric_amd import _native
    from tests._util import decompress_rows
    n = 517
    x = _with_zeros(n, F, 0.5, F)
    x[3] = 0                      # an empty row
    x[4] = torch.arange(1, F + 1)  # a full row
    x[5, 0] = -0.0
    x[5, F - 1] = float('nan')
    x[6] = 0
    x[6, F - 1] = 1e-42
    wide = torch.zeros(n, F + 5)
    wide[:, 1:1 + F] = x
    z = _native.rows_compress(wide.to(dev)[:, 1:1 + F])  # a strided, unaligned source
    assert z.dtype == torch.int32 and z.shape == (n, _native.compressed_pitch(F))
    got, mask = decompress_rows(z, F)
    want_mask = x.view(torch.int32) != 0
    assert torch.equal(mask, want_mask)
    assert torch.equal(got.view(torch.int32), x.view(torch.int32))
    # the mask words past F are clear
    import numpy as np
    zz = z.cpu().numpy().view(np.uint32)
    for c in range(F, 256):
        assert not ((zz[:, c >> 5] >> np.uint32(c & 31)) & 1).any()
    with pytest.raises(ValueError):
        _native.rows_compress(torch.zeros(4, 260, device=dev))
    assert _native.rows_compress(torch.zeros(0, F, device=dev)).shape[0] == 0


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('F', [4, 64, 100, 132, 256])
def test_spmm_from_compressed_rows_is_bit_exact(dev, dtype, F, monkeypatch):
    """The aggregation reads compressed source rows (x_format = PYGAMD_X_COMPRESSED).  A whole wave
    decodes one source row, slot after slot: for F > 128, where the dense kernel walks the slots the
    same way, the sums are the same bit for bit (same values, same order; a dropped +0.0 changes no
    sum); narrower rows (the dense kernel then adds several slots side by side) agree to fp32
    rounding.  Hub rows and rows with more than 64 / 8 k + r slots included."""
    from pytorch_geometric_amd import _native
    import pytorch_geometric_amd as pga
    monkeypatch.setattr(_native, 'HUB_THRESHOLD', 200)
    monkeypatch.setattr(_native, 'HUB_CHUNK', 96)
    orig = _native.hub_plan
    monkeypatch.setattr(_native, 'hub_plan', lambda ptr, threshold=None, chunk=None: orig(
        ptr, 200, 96))
    n_src, n_dst = 2000, 700
    ei = random_graph(n_src, n_dst, 30_000, seed=F + 11, dtype=dtype, skew=True)
    h = pga.EdgeIndex(ei.to(dev), (n_src, n_dst)).by_dst()
    assert h.hub[2] > 0
    for density in (0.5, 0.0, 1.0, 0.03):
        x = _with_zeros(n_src, F, density, F + int(density * 100)).to(dev)
        z = _native.rows_compress(x)
        for red in ('sum', 'mean'):
            dense = _native.spmm_csr(h.ptr, h.idx, x, red, n_rows=n_dst, hub=h.hub)
            out = _native.spmm_csr(h.ptr, h.idx, z, red, n_rows=n_dst, hub=h.hub,
                                   compressed_width=F)
            if F > 128:
                assert torch.equal(out.view(torch.int32), dense.view(torch.int32)), \
                    f'{red} F={F} density={density}: {(out - dense).abs().max().item():.3e}'
            else:
                ex = O.spmm(ei.long(), x.cpu().double(), n_dst, red)
                assert_sum_close(out, dense, ex, what=f'compressed {red} F={F} p={density}')
    with pytest.raises(ValueError):
        _native.spmm_csr(h.ptr, h.idx, z, 'max', n_rows=n_dst, hub=h.hub, compressed_width=F)
    with pytest.raises(ValueError):
        _native.spmm_csr(h.ptr, h.idx, z[:, :F], 'sum', n_rows=n_dst, hub=h.hub,
                         compressed_width=F)


def test_large_unsorted_scatter_takes_the_sorted_route(dev, monkeypatch):
    """`utils.scatter` on a large unsorted index: one cached radix sort + a segment reduction
    instead of E x F float atomics (the literal north_star path at the products shape: 206 ms ->
    sort + 11.5 ms, profiles/r04_unfused_propagate.md).  Same values as the atomic kernels
    (max / min exactly, sums to fp32 rounding) and as the CPU, same gradients; the plan is cached per
    index tensor — also through fresh views like `edge_index[1]` — and dropped when the index is
    edited in place.  Route rule: an index below SORTED_SCATTER_ALWAYS_ROWS entries
    is sorted from its SECOND sighting on (a sampled batch's one-shot index keeps the atomics), a
    row of an `EdgeIndex` handle or a larger index at once; building a plan reads nothing back to
    the host (`pygamd_index_guard`): out-of-range entries land in a sentinel group that is skipped,
    and the flag travels as PYGAMD_CHECK_INDEX says; plans are int32 and the cache is bounded in
    bytes."""
    from pytorch_geometric_amd import _functions, _native
    from pytorch_geometric_amd.utils import scatter
    g = gen(77)
    n, e, F = 5000, 140_000, 64
    ei = torch.stack([torch.randint(0, n, (e, ), generator=g),
                      (torch.rand(e, generator=g).pow(3) * n).long()]).to(dev)
    src = torch.randn(e, F, generator=g).to(dev)
    assert _functions._use_sorted_scatter(src, ei[1], 'sum')
    sorts = {'n': 0}
    real = _native.index_sort

    def counted(*a, **k):
        sorts['n'] += 1
        return real(*a, **k)

    monkeypatch.setattr(_native, 'index_sort', counted)
    minmax = {'n': 0}
    real_mm = _native.index_minmax

    def counted_mm(*a, **k):
        minmax['n'] += 1
        return real_mm(*a, **k)

    monkeypatch.setattr(_native, 'index_minmax', counted_mm)
    _functions._scatter_plans.clear()
    _functions._scatter_seen.clear()
    assert e < _functions.SORTED_SCATTER_ALWAYS_ROWS
    scatter(src, ei[1], 0, n, 'sum')     # first sighting: atomics, nothing sorted, nothing cached
    assert sorts['n'] == 0 and not _functions._scatter_plans and len(_functions._scatter_seen) == 1
    for reduce in ('sum', 'mean', 'max', 'min'):
        s = src.clone().requires_grad_(True)
        out = scatter(s, ei[1], 0, n, reduce)
        go = torch.randn(n, F, generator=g).to(dev)
        out.backward(go)
        want_s = src.cpu().clone().requires_grad_(True)
        want = torch.zeros(n, F).scatter_reduce(0, ei[1].cpu().view(-1, 1).expand(-1, F), want_s,
                                                {'sum': 'sum', 'mean': 'mean', 'max': 'amax',
                                                 'min': 'amin'}[reduce], include_self=False)
        want.backward(go.cpu())
        # (the skewed index makes groups of thousands of rows: sums are judged against fp64)
        ex = torch.zeros(n, F, dtype=torch.float64).scatter_reduce(
            0, ei[1].cpu().view(-1, 1).expand(-1, F), src.cpu().double(),
            {'sum': 'sum', 'mean': 'mean', 'max': 'amax', 'min': 'amin'}[reduce],
            include_self=False)
        assert_sum_close(out, want.detach(), ex, what=f'sorted scatter {reduce}')
        assert_close(s.grad, want_s.grad, rtol=1e-5, atol=2e-5, what=f'sorted scatter {reduce} grad')
        atom = _native.scatter_rows(src, ei[1], n, reduce)
        if reduce in ('max', 'min'):
            assert torch.equal(out.detach(), atom)
        else:
            assert_sum_close(atom, want.detach(), ex, what='atomic route, same inputs')
    assert sorts['n'] == 1, 'the sort must be cached across calls and across views of the index'
    assert minmax['n'] == 0, 'building or using a plan must not read the index range back'
    (entry, ) = _functions._scatter_plans.values()
    ptr, perm, hub = entry[2]
    assert ptr.dtype == perm.dtype == torch.int32 and ptr.numel() == n + 1 and perm.numel() == e
    assert entry[4] and hub is not None, 'the hub plan is added when a plan is reused'
    assert entry[3] == 4 * (n + 1 + e)
    ei[1, 0] = (ei[1, 0] + 1) % n          # in-place edit: the cached plan is stale
    scatter(src, ei[1], 0, n, 'sum')       # ... and this version of the index is new: atomics
    assert sorts['n'] == 1
    scatter(src, ei[1], 0, n, 'sum')
    assert sorts['n'] == 2
    # a row of an EdgeIndex handle, or an index past the always-threshold: sorted at first sight
    from pytorch_geometric_amd.edge_index import EdgeIndex
    h = EdgeIndex(ei.clone(), sparse_size=(n, n), validate=False)
    scatter(src, h[1], 0, n, 'sum')
    assert sorts['n'] == 3
    monkeypatch.setattr(_functions, 'SORTED_SCATTER_ALWAYS_ROWS', 1 << 16)
    fresh = ei[1].clone()
    scatter(src, fresh, 0, n, 'sum')
    assert sorts['n'] == 4
    # out-of-range entries: skipped like the atomic kernels skip them, flagged without a wait
    import pytorch_geometric_amd as pga
    pga.check_index_errors()
    bad = fresh.clone()
    bad[5], bad[77] = n + 3, -1
    keep = torch.ones(e, dtype=torch.bool, device=dev)
    keep[5] = keep[77] = False
    want = _native.scatter_rows(src[keep], fresh[keep], n, 'sum')
    got = scatter(src, bad, 0, n, 'sum')
    assert sorts['n'] == 5 and minmax['n'] == 0
    # (groups of thousands of rows, summed in another order: judged at the size of the sums)
    assert_close(got, want, rtol=1e-4, atol=2e-3, what='out-of-range rows are skipped')
    with pytest.raises(IndexError, match='out of bounds'):
        pga.check_index_errors()
    pga.check_index_errors()
    # the plan of `bad` is cached: a LATER use of the same tensor reports its entries again, at
    # the call and without device work (only the first use used to say so)
    with pytest.raises(IndexError, match='out of bounds'):
        scatter(src, bad, 0, n, 'sum')
    assert sorts['n'] == 5
    pga.check_index_errors()
    _native.INDEX_CHECK = 'sync'
    try:
        with pytest.raises(IndexError, match='out of bounds'):
            scatter(src, bad.clone(), 0, n, 'sum')
    finally:
        _native.INDEX_CHECK = 'async'
    pga.check_index_errors()
    # the cache is bounded in bytes: with room for one plan the older one goes
    _functions._scatter_plans.clear()
    monkeypatch.setattr(_functions, 'SCATTER_PLAN_CACHE_BYTES', 4 * (n + 1 + e) + 64)
    a, b = fresh.clone(), fresh.clone()
    scatter(src, a, 0, n, 'sum')
    scatter(src, b, 0, n, 'sum')
    assert len(_functions._scatter_plans) == 1
    assert next(iter(_functions._scatter_plans))[0] == id(b)


@pytest.mark.parametrize('H,C', [(8, 32), (8, 40), (1, 16), (4, 4), (2, 6), (1, 5)])
@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_sddmm_spmm_one_gather(dev, H, C, dtype):
    """`pygamd_sddmm_spmm_csr`: the SDDMM (per-edge, per-head dot products) and the per-head
    weighted aggregation over the same CSR from ONE gather — the GAT backward over the by-source
    form.  Against the two stand-alone entry points (same kernels' arithmetic: dot products
    bitwise, sums to rounding) and against fp64; widths that take the 16-byte lanes, two chunks
    per lane (F = 320: heads spanning lane groups, atomic adds), several rows per wave (F = 16) and
    the scalar lanes (C = 5, 6); an edge-id map (results filed under another slot order)."""
    from pytorch_geometric_amd import _native
    g = gen(H * 100 + C)
    n_src, n_dst, e = 700, 500, 9000
    F = H * C
    src = torch.randint(0, n_src, (e, ), generator=g)
    src[:400] = 3                                    # one long row
    dst = torch.randint(0, n_dst, (e, ), generator=g)
    order = torch.argsort(src, stable=True)
    src, dst = src[order], dst[order]
    ptr = torch.zeros(n_src + 1, dtype=torch.long)
    ptr[1:] = torch.bincount(src, minlength=n_src).cumsum(0)
    eid = torch.randperm(e, generator=g)             # where edge k's weight / result lives
    rows = torch.randn(n_src, F, generator=g)
    x = torch.randn(n_dst, F, generator=g)
    w = torch.rand(e, H, generator=g)
    gw, agg = _native.sddmm_spmm_csr(ptr.to(dtype).to(dev), dst.to(dtype).to(dev),
                                     eid.to(dtype).to(dev), rows.to(dev), x.to(dev), w.to(dev),
                                     e, H)
    # fp64 references
    xr = x[dst].double().view(e, H, C)
    ex_gw = torch.zeros(e, H, dtype=torch.float64)
    ex_gw[eid] = (rows[src].double().view(e, H, C) * xr).sum(-1)
    ex_agg = torch.zeros(n_src, H, C, dtype=torch.float64).index_add_(
        0, src, w[eid].double().unsqueeze(-1) * xr).view(n_src, F)
    bound_gw = torch.zeros(e, H, dtype=torch.float64)
    bound_gw[eid] = (rows[src].double().view(e, H, C).abs() * xr.abs()).sum(-1)
    bound_agg = torch.zeros(n_src, H, C, dtype=torch.float64).index_add_(
        0, src, w[eid].double().unsqueeze(-1) * xr.abs()).view(n_src, F)
    assert bool(((gw.cpu().double() - ex_gw).abs() <= 1e-5 * bound_gw + 1e-12).all())
    assert bool(((agg.cpu().double() - ex_agg).abs() <= 1e-5 * bound_agg + 1e-12).all())
    # the stand-alone entry points on the same inputs
    gw2 = _native.sddmm_csr(ptr.to(dtype).to(dev), dst.to(dtype).to(dev), eid.to(dtype).to(dev),
                            rows.to(dev), x.to(dev), e, H)
    assert_close(gw, gw2, rtol=1e-6, atol=1e-6, what='fused vs stand-alone SDDMM')
    agg2 = _native.spmm_csr(ptr.to(dtype).to(dev), dst.to(dtype).to(dev), x.to(dev), 'sum',
                            n_rows=n_src, eid=eid.to(dtype).to(dev), w=w.to(dev))
    assert_close(agg, agg2, rtol=1e-5, atol=1e-5, what='fused vs stand-alone weighted SpMM')
    # strided operands (columns of wider buffers, 16-byte aligned or not)
    for off in (4, 3):
        big_r = torch.randn(n_src, F + 8, generator=g).to(dev)
        big_x = torch.randn(n_dst, F + 8, generator=g).to(dev)
        gw3, agg3 = _native.sddmm_spmm_csr(
            ptr.to(dtype).to(dev), dst.to(dtype).to(dev), None, big_r[:, off:off + F],
            big_x[:, off:off + F], w.to(dev), e, H)
        gw4 = _native.sddmm_csr(ptr.to(dtype).to(dev), dst.to(dtype).to(dev), None,
                                big_r[:, off:off + F].contiguous(),
                                big_x[:, off:off + F].contiguous(), e, H)
        assert_close(gw3, gw4, rtol=1e-5, atol=1e-5, what=f'strided SDDMM off={off}')
        agg4 = _native.spmm_csr(ptr.to(dtype).to(dev), dst.to(dtype).to(dev),
                                big_x[:, off:off + F].contiguous(), 'sum', n_rows=n_src,
                                w=w.to(dev))
        assert_close(agg3, agg4, rtol=1e-5, atol=1e-5, what=f'strided SpMM off={off}')


@pytest.mark.parametrize('F', [1, 5, 48, 100, 256, 320])
@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_spmm_minmax_without_argument(dev, F, dtype):
    """`spmm_minmax_rows_plain`: the extremum alone (no argument / tie output requested — what the
    unfused `scatter(..., 'max')` path and inference take): bit-equal to the argument-tracking
    kernel and to ATen's amax / amin, including NaN (propagates), +-inf data, an all -inf row, empty
    rows (-> 0), hub rows and the identity-column form (`col = None`: a segment reduction)."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    n_src, n_dst = 400, 330
    ei = random_graph(n_src, n_dst, 8000, seed=F + 5, skew=True)
    ei[1][ei[1] == 7] = 8                       # destination 7 is empty
    extra = torch.stack([torch.full((700, ), 3), torch.full((700, ), 11)])   # a 700-slot row
    ei = torch.cat([ei, extra], dim=1)
    g = gen(F + 31)
    x = torch.randn(n_src, F, generator=g)
    x[5] = float('nan')
    x[6, ::2] = float('inf')
    x[9] = float('-inf')
    only = (ei[0] == 9).nonzero().view(-1)
    if only.numel() > 0:                        # a destination whose every neighbour is row 9
        ei[1, only] = 12
        ei = ei[:, ~((ei[1] == 12) & (ei[0] != 9))]
    h = pga.EdgeIndex(ei.to(dtype).to(dev), (n_src, n_dst))
    fwd = h.by_dst()
    xg = x.to(dev)
    idx = ei[1].view(-1, 1).expand(-1, F)
    for red, aten in (('max', 'amax'), ('min', 'amin')):
        want = torch.zeros(n_dst, F).scatter_reduce(0, idx, x[ei[0]], aten, include_self=False)
        got = _native.spmm_csr(fwd.ptr, fwd.idx, xg, red, n_rows=n_dst, hub=fwd.hub)
        assert torch.equal(torch.nan_to_num(got.cpu(), nan=123.0),
                           torch.nan_to_num(want, nan=123.0)), red
        assert bool(torch.isnan(got.cpu()).eq(torch.isnan(want)).all())
        tracked, _ = _native.spmm_csr(fwd.ptr, fwd.idx, xg, red, n_rows=n_dst, hub=fwd.hub,
                                      save_arg32=True)
        assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(tracked, nan=123.0))
        # identity columns: rows grouped by pointer (utils.segment's reduction)
        msgs = xg[fwd.idx.long()]
        seg = _native.spmm_csr(fwd.ptr, None, msgs, red, n_rows=n_dst, hub=fwd.hub)
        assert torch.equal(torch.nan_to_num(seg, nan=123.0), torch.nan_to_num(got, nan=123.0))

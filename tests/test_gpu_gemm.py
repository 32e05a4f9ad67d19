"""The fp32-MFMA dense transform (csrc/gemm.hip) against torch's CPU GEMM: exact-fp32 MFMA is an
fmaf chain, so the 1e-5 contract holds relative to the size of the dot product's terms
(`assert_sum_close` with the |a|.|b| bound — K up to thousands of terms, wgrad sums over M rows)."""
import pytest
import torch

from tests._util import assert_close, assert_sum_close, gen

pytestmark = pytest.mark.gpu

SHAPES = [  # (M, K, N)
    (1, 1, 1), (5, 3, 2), (33, 7, 31), (64, 32, 32), (130, 200, 256), (257, 512, 256),
    (300, 256, 96), (129, 100, 47), (1000, 36, 130), (70, 16, 7), (513, 1433, 16), (40, 0, 8),
    (0, 16, 8),
]


def _fwd_ref(x, w, b, relu):
    out = x.double() @ w.double().t()
    if b is not None:
        out = out + b.double()
    return out.relu() if relu else out


@pytest.mark.parametrize('M,K,N', SHAPES)
def test_linear_forward(dev, M, K, N):
    from pytorch_geometric_amd import _native
    g = gen(M * 31 + K * 7 + N)
    x, w, b = (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g),
               torch.randn(N, generator=g))
    for bias, relu in ((None, False), (b, False), (b, True)):
        ex = _fwd_ref(x, w, bias, relu)
        ref32 = torch.nn.functional.linear(x, w, bias)
        ref32 = ref32.relu() if relu else ref32
        bound = x.abs().double() @ w.abs().double().t() + (0 if bias is None else bias.abs())
        out = _native.linear_forward(x.to(dev), w.to(dev), None if bias is None else bias.to(dev),
                                     relu=relu)
        assert out.shape == (M, N)
        assert_sum_close(out, ref32, ex, abs_sum=bound, what=f'fwd {M}x{K}x{N} relu={relu}')


def test_linear_forward_strided_views_and_accumulate(dev):
    """Operands and outputs as halves of wider buffers (the `[agg | x]` layout), unaligned
    (scalar-load) fallbacks, accumulate."""
    from pytorch_geometric_amd import _native
    g = gen(5)
    big = torch.randn(300, 520, generator=g).to(dev)
    w = torch.randn(96, 256, generator=g).to(dev)
    b = torch.randn(96, generator=g).to(dev)
    outbuf = torch.zeros(300, 200, device=dev)
    for off in (0, 256, 3):  # 3: rows no longer 16-byte aligned -> scalar loads
        xv = big[:, off:off + 256]
        ov = outbuf[:, 100:196]
        _native.linear_forward(xv, w, b, relu=True, out=ov)
        ref = torch.nn.functional.linear(xv.cpu(), w.cpu(), b.cpu()).relu()
        assert_close(ov, ref, rtol=1e-5, atol=2e-4, what=f'strided off={off}')
        assert bool((outbuf[:, :100] == 0).all()) and bool((outbuf[:, 196:] == 0).all())
    base = torch.randn(300, 96, generator=g).to(dev)
    acc = base.clone()
    _native.linear_forward(big[:, :256], w, None, out=acc, accumulate=True)
    ref = base.cpu() + big[:, :256].cpu() @ w.cpu().t()
    assert_close(acc, ref, rtol=1e-5, atol=2e-4, what='accumulate')


@pytest.mark.parametrize('M,K,N', SHAPES)
def test_linear_dgrad_and_wgrad(dev, M, K, N):
    from pytorch_geometric_amd import _native
    g = gen(M + K * 13 + N * 101)
    x, w, go = (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g),
                torch.randn(M, N, generator=g))
    scale = torch.rand(M, generator=g) + 0.5
    # dgrad: go @ w, left `ns` columns scaled per row
    ns = K // 2
    ex = go.double() @ w.double()
    ex[:, :ns] *= scale.double().view(-1, 1)
    ref32 = go @ w
    ref32[:, :ns] *= scale.view(-1, 1)
    bound = go.abs().double() @ w.abs().double()
    bound[:, :ns] *= scale.double().view(-1, 1)
    out = _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), scale.to(dev), ns)
    assert out.shape == (M, K)
    assert_sum_close(out, ref32, ex, abs_sum=bound, what=f'dgrad {M}x{K}x{N}')
    # wgrad: go^T @ x
    ex = go.double().t() @ x.double()
    ref32 = go.t() @ x
    bound = go.abs().double().t() @ x.abs().double()
    out = _native.linear_wgrad(go.to(dev), x.to(dev))
    assert out.shape == (N, K)
    assert_sum_close(out, ref32, ex, abs_sum=bound, what=f'wgrad {M}x{K}x{N}')
    # the bias gradient from the same pass over `go`; the weight gradient is bit-identical
    out2, gb = _native.linear_wgrad(go.to(dev), x.to(dev), bias_grad=True)
    assert torch.equal(out2, out) and gb.shape == (N, )
    assert_sum_close(gb, go.sum(0), go.double().sum(0), abs_sum=go.abs().double().sum(0),
                     what=f'wgrad bias {M}x{K}x{N}')
    # two operands side by side: g^T @ [x | x2] without the concatenated copy (strided x2: the
    # right half of a wider buffer); the number of row splits differs from the one-operand launch,
    # so the comparison is against fp64 like everywhere else
    if K > 0:
        K2 = K // 2 + 3
        wide = torch.randn(M, 2 * K2, generator=g)
        x2 = wide[:, K2:]
        both = torch.cat([x, x2], 1)
        got, gb2 = _native.linear_wgrad(go.to(dev), x.to(dev), bias_grad=True,
                                        x2=wide.to(dev)[:, K2:])
        assert got.shape == (N, K + K2)
        assert_sum_close(got, go.t() @ both, go.double().t() @ both.double(),
                         abs_sum=go.abs().double().t() @ both.abs().double(),
                         what=f'wgrad two operands {M}x({K}+{K2})x{N}')
        assert_sum_close(gb2, go.sum(0), go.double().sum(0), abs_sum=go.abs().double().sum(0),
                         what='wgrad two operands, bias')
    # dgrad with the ReLU-backward epilogue: exact zeros where the mask is not positive, the plain
    # result (bit-identical) elsewhere; strided mask (right half of an [agg | x] buffer)
    if K > 0:
        act = torch.randn(M, 2 * K, generator=g).relu()
        act[::7, K:] = -0.0
        act_d = act.to(dev)
        plain = _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), scale.to(dev), ns)
        got = _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), scale.to(dev), ns,
                                   relu_mask=act_d[:, K:])
        assert torch.equal(got, torch.where(act_d[:, K:] > 0, plain, torch.zeros_like(plain)))
        # the same mask as one bit per element (rows padded to one more word than needed)
        packed = _native.pack_relu_bits(act_d[:, K:])
        wide = torch.zeros(packed.size(0), packed.size(1) + 1, 32, dtype=torch.int32, device=dev)
        wide[:, :packed.size(1)] = packed
        got_b = _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), scale.to(dev), ns,
                                     relu_bits=wide)
        assert torch.equal(got_b, got)
        base = torch.randn(M, K, generator=g).to(dev)
        acc = base.clone()
        _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), out=acc, accumulate=True,
                             relu_mask=act_d[:, K:])
        want = torch.where(act_d[:, K:] > 0,
                           base + _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev)),
                           torch.zeros_like(base))
        assert_close(acc, want, rtol=1e-6, atol=1e-6, what='dgrad accumulate + mask')
        acc_b = base.clone()
        _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), out=acc_b, accumulate=True,
                             relu_bits=packed)
        assert torch.equal(acc_b, acc)


def test_wgrad_long_reduction_is_deterministic_and_accurate(dev):
    """M = 200k rows split over many workgroups: slabs are summed in split order (no atomics), so
    two runs are bitwise equal; accuracy judged against fp64."""
    from pytorch_geometric_amd import _native
    g = gen(9)
    M, N, K = 200_003, 96, 200
    go, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    a, ba = _native.linear_wgrad(go.to(dev), x.to(dev), bias_grad=True)
    b, bb = _native.linear_wgrad(go.to(dev), x.to(dev), bias_grad=True)
    assert torch.equal(a, b) and torch.equal(ba, bb)
    assert_sum_close(ba, go.sum(0), go.double().sum(0), abs_sum=go.abs().double().sum(0),
                     what='wgrad bias M=200003')
    ex = go.double().t() @ x.double()
    bound = go.abs().double().t() @ x.abs().double()
    assert_sum_close(a, go.t() @ x, ex, abs_sum=bound, what='wgrad M=200003')
    acc = torch.ones(N, K, device=dev)
    _native.linear_wgrad(go.to(dev), x.to(dev), out=acc, accumulate=True)
    assert_sum_close(acc - 1, go.t() @ x, ex, abs_sum=bound + 1, what='wgrad accumulate')


@pytest.mark.parametrize('F,Fo,reduce', [(256, 256, 'mean'), (100, 256, 'mean'), (64, 200, 'sum'),
                                         (8, 47, 'mean'), (128, 32, 'sum')])
@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('variant', [1, 2, 3, 4])  # row-at-a-time / streamed gather phase /
# producer-consumer waves with 4 or 8 transform waves
def test_sage_layer_forward_one_kernel(dev, F, Fo, reduce, dtype, variant):
    """csrc/sage_fused.hip: aggregation + transform + bias + ReLU of a SAGEConv layer in one
    kernel against the oracle's sage_conv (index_select + scatter + two matmuls), with hub rows,
    empty rows, a row count that is no multiple of the 32-row tile, and strided operands (the
    `[agg | x]` buffer layout); the saved aggregated rows are checked too."""
    import pytorch_geometric_amd as pga
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd import _native
    from tests._util import random_graph
    n = 1037
    g = gen(F * 3 + Fo)
    ei = random_graph(n, n, 30000, seed=F + Fo, skew=True)   # a few hub destinations
    ei[1][ei[1] == 5] = 6                                      # row 5 has no in-edges
    x = torch.randn(n, F, generator=g)
    wl, wr = torch.randn(Fo, F, generator=g) * 0.1, torch.randn(Fo, F, generator=g) * 0.1
    b = torch.randn(Fo, generator=g)
    aggr_ref = O.spmm(ei, x, n, reduce)
    ref = (aggr_ref @ wl.t() + x @ wr.t() + b).relu()
    ex = (aggr_ref.double() @ wl.double().t() + x.double() @ wr.double().t() + b.double()).relu()
    # rounding scales with the sum of |terms|: hub rows aggregate thousands of |x_j|
    aggr_abs = O.spmm(ei, x.abs(), n, reduce).double()
    bound = aggr_abs @ wl.abs().double().t() + x.abs().double() @ wr.abs().double().t()
    h = pga.EdgeIndex(ei.to(dtype).to(dev), (n, n))
    fwd = h.by_dst()
    assert fwd.hub[2] > 0
    buf = torch.full((n, 2 * F), float('nan'), device=dev)
    buf[:, F:] = x.to(dev)
    out = torch.full((n, Fo + 8), float('nan'), device=dev)
    wcat = torch.cat([wl, wr], 1).to(dev)
    assert _native.sage_layer_forward_supported(F, Fo, reduce)
    bits = torch.full(((n + 31) // 32, (Fo + 31) // 32 + 1, 32), -1, dtype=torch.int32,
                      device=dev)
    _native.sage_layer_forward(fwd.ptr, fwd.idx, x.to(dev), buf[:, F:], wcat, b.to(dev), reduce,
                               True, buf[:, :F], out[:, :Fo], hub=fwd.hub, save_agg=True,
                               relu_bits=bits, variant=variant)
    assert_sum_close(out[:, :Fo], ref, ex, abs_sum=bound + 1, what=f'fused layer F={F} Fo={Fo}')
    # the one-bit-per-element ReLU mask written next to the output: exactly [out > 0], zero bits
    # past Fo inside the last word, nothing written past the last word
    want_bits = _native.pack_relu_bits(out[:, :Fo])
    tail = n - (n // 32) * 32  # rows of the last (partial) tile that exist
    assert torch.equal(bits[:-1, :-1], want_bits[:-1])
    assert torch.equal(bits[-1, :-1, :tail], want_bits[-1, :, :tail])
    assert bool((bits[-1, :-1, tail:] == -1).all())  # rows past n_rows: untouched
    assert bool((bits[:, -1] == -1).all())            # column blocks past Fo: untouched
    assert 0 < int((out[:, :Fo] > 0).sum()) < n * Fo
    assert bool(torch.isnan(out[:, Fo:]).all())               # nothing written past Fo
    assert_sum_close(buf[:, :F], aggr_ref, O.spmm(ei, x.double(), n, reduce),
                     what='saved aggregated rows')
    # without ReLU / bias, gathering from the strided half of the buffer itself
    out2 = torch.empty(n, Fo, device=dev)
    _native.sage_layer_forward(fwd.ptr, fwd.idx, buf[:, F:], buf[:, F:], wcat, None, reduce, False,
                               buf[:, :F], out2, hub=fwd.hub, save_agg=False, variant=variant)
    ref2 = aggr_ref @ wl.t() + x @ wr.t()
    ex2 = aggr_ref.double() @ wl.double().t() + x.double() @ wr.double().t()
    assert_sum_close(out2, ref2, ex2, abs_sum=bound + 1, what='fused layer, no bias / relu')


@pytest.mark.parametrize('F,Fo', [(256, 256), (100, 256), (48, 64), (8, 47)])
def test_sage_layer_streamed_gather_is_bitwise_the_spmm(dev, F, Fo):
    """The streamed gather phase (variant 2) adds a row's slots in the order pygamd_spmm_csr does:
    output, saved aggregated rows and ReLU bits are bitwise those of the row-at-a-time variant
    (itself bitwise SpMM + GEMM).  The graph has hub rows, empty rows, a tile whose non-hub slots
    exceed the LDS index cache (indices then come from global memory) and a partial last tile."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    from tests._util import random_graph
    n = 1500
    g = gen(F + 7 * Fo)
    ei = random_graph(n, n, 40000, seed=3 * F + Fo, skew=True)
    # rows 64..95 (one tile): 200 in-edges each = 6400 slots > the 3072 staged indices
    dense_dst = torch.arange(64, 96).repeat_interleave(200)
    dense_src = torch.randint(0, n, (dense_dst.numel(), ), generator=g)
    ei = torch.cat([ei, torch.stack([dense_src, dense_dst])], 1)
    ei = ei[:, (ei[1] != 5) & (ei[1] != 130) & (ei[1] != 131)]  # rows without in-edges
    x = torch.randn(n, F, generator=g).to(dev)
    w = (torch.randn(Fo, 2 * F, generator=g) * 0.1).to(dev)
    b = torch.randn(Fo, generator=g).to(dev)
    h = pga.EdgeIndex(ei.to(dev), (n, n))
    fwd = h.by_dst()
    assert fwd.hub[2] > 0
    res = []
    for variant in (1, 2):
        agg = torch.full((n, F), float('nan'), device=dev)
        out = torch.full((n, Fo), float('nan'), device=dev)
        bits = _native.relu_bits_like(n, Fo, dev).fill_(-1)
        _native.sage_layer_forward(fwd.ptr, fwd.idx, x, x, w, b, 'mean', True, agg, out,
                                   hub=fwd.hub, save_agg=True, relu_bits=bits, variant=variant)
        res.append((agg, out, bits))
    for a, b2, what in zip(res[0], res[1], ('aggregated rows', 'output', 'ReLU bits')):
        assert torch.equal(a, b2), f'{what}: streamed variant differs from row-at-a-time'
    two = _native.spmm_csr(fwd.ptr, fwd.idx, x, 'mean', n_rows=n, hub=fwd.hub)
    assert torch.equal(res[1][0], two), 'aggregated rows differ from pygamd_spmm_csr'
    assert not bool(torch.isnan(res[1][1]).any())


@pytest.mark.parametrize('F,Fo', [(256, 256), (64, 128), (100, 96), (32, 32), (192, 64)])
@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_sage_layer_with_compressed_rows_in_and_out(dev, request, F, Fo, dtype):
    """The one-kernel layer gathers COMPRESSED source rows (x_format) and writes its own output a
    second time in that layout (compressed_out): for F > 128 bitwise the dense launch — output,
    saved aggregated rows, ReLU bits (narrower rows: equal to fp32 rounding, the dense gather then
    adds several slots side by side) — and compressed_out decodes to exactly the launch's own dense
    output.  Hub rows, empty rows, a partial last tile; the source is a ReLU output (half
    zeros)."""
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd import _native
    from tests._util import assert_close_scaled, decompress_rows, random_graph
    # (compressed rows in / out run the fp32-instruction schedule whatever the arithmetic mode: the
    # bitwise comparisons below need the dense launches on the same schedule)
    request.addfinalizer(lambda prev=_native.set_gemm_mode('fp32'): _native.set_gemm_mode(prev))
    n = 1037
    g = gen(5 * F + Fo)
    ei = random_graph(n, n, 30000, seed=F + 2 * Fo, skew=True).to(dtype)
    ei = ei[:, (ei[1] != 5) & (ei[1] != 77)]
    x = torch.randn(n, F, generator=g).relu().to(dev)
    x[9] = 0
    w = (torch.randn(Fo, 2 * F, generator=g) * 0.1).to(dev)
    b = torch.randn(Fo, generator=g).to(dev)
    fwd = pga.EdgeIndex(ei.to(dev), (n, n)).by_dst()
    assert fwd.hub[2] > 0
    z = _native.rows_compress(x)
    res = []
    for zin in (None, z):
        agg = torch.full((n, F), float('nan'), device=dev)
        out = torch.full((n, Fo), float('nan'), device=dev)
        bits = _native.relu_bits_like(n, Fo, dev).fill_(-1)
        zout = torch.full((n, _native.compressed_pitch(Fo)), -1, dtype=torch.int32, device=dev)
        _native.sage_layer_forward(fwd.ptr, fwd.idx, x if zin is None else zin, x, w, b, 'mean',
                                   True, agg, out, hub=fwd.hub, save_agg=True, relu_bits=bits,
                                   gather_width=None if zin is None else F,
                                   compressed_out=zout)
        res.append((agg, out, bits))
        dec, mask = decompress_rows(zout, Fo)
        assert torch.equal(dec.view(torch.int32), out.cpu().view(torch.int32)), \
            'compressed_out does not decode to the dense output'
        assert torch.equal(mask, out.cpu() > 0)
    for a, c, what in zip(res[0], res[1], ('aggregated rows', 'output', 'ReLU bits')):
        if F > 128:
            assert torch.equal(a, c), f'{what}: compressed gather differs from the dense gather'
        elif what != 'ReLU bits':
            assert_close_scaled(a, c, what=what)
    assert not bool(torch.isnan(res[1][1]).any()) and 0 < int((res[1][1] > 0).sum())
    # no ReLU, no bias, a signed output (everything kept but exact zeros), compressed in only
    out_d, out_z = torch.empty(n, Fo, device=dev), torch.empty(n, Fo, device=dev)
    agg = torch.empty(n, F, device=dev)
    _native.sage_layer_forward(fwd.ptr, fwd.idx, x, x, w, None, 'sum', False, agg, out_d,
                               hub=fwd.hub, save_agg=False)
    _native.sage_layer_forward(fwd.ptr, fwd.idx, z, x, w, None, 'sum', False, agg, out_z,
                               hub=fwd.hub, save_agg=False, gather_width=F)
    if F > 128:
        assert torch.equal(out_d, out_z)
    else:
        assert_close_scaled(out_z, out_d, what='signed output')
    with pytest.raises(Exception):  # the other schedules do not take compressed rows
        _native.sage_layer_forward(fwd.ptr, fwd.idx, z, x, w, None, 'sum', False, agg, out_z,
                                   hub=fwd.hub, save_agg=False, gather_width=F, variant=3)


@pytest.mark.parametrize('Fi,Fo', [(256, 256), (64, 128), (100, 40)])
@pytest.mark.parametrize('reduce', ['mean', 'sum'])
@pytest.mark.parametrize('variant', [1, 2, 3, 4])
def test_sage_layer_input_gradient_one_kernel(dev, Fi, Fo, reduce, variant):
    """pygamd_sage_layer_fused as a layer's INPUT GRADIENT: on the transposed graph, with the
    1/deg-scaled gradient rows gathered, the unscaled ones as root operand, w = [W_l^T | W_r^T] and
    the ReLU mask of the layer input as bits, one launch equals the oracle's autograd through
    sage_conv(relu(h)) — and the second, row-scaled output is that times row_scale."""
    import pytorch_geometric_amd as pga
    from oracle import pyg_oracle as O
    from pytorch_geometric_amd import _native
    from tests._util import random_graph
    n = 1100
    g = gen(Fi + Fo + len(reduce))
    ei = random_graph(n, n, 30000, seed=Fi * 3 + Fo, skew=True)
    ei[0][ei[0] == 9] = 10                           # node 9 has no out-edges (empty transposed row)
    # hubs in the TRANSPOSED graph too: a few very popular sources
    ei[0][:3000] = torch.randint(0, 2, (3000, ), generator=g)
    pre = torch.randn(n, Fi, generator=g)            # pre-activation of the layer below
    wl, wr = torch.randn(Fo, Fi, generator=g) * 0.1, torch.randn(Fo, Fi, generator=g) * 0.1
    go = torch.randn(n, Fo, generator=g)
    pr = pre.clone().requires_grad_(True)
    out = O.sage_conv(pr.relu(), ei, wl, None, wr, reduce)
    out.backward(go)
    ref = pr.grad
    h = pga.EdgeIndex(ei.to(dev), (n, n))
    bwd, fwd = h.by_src(), h.by_dst()
    assert bwd.hub[2] > 0
    scale = fwd.inv_degree() if reduce == 'mean' else None
    g_dev = go.to(dev)
    gs = g_dev * scale.view(-1, 1) if scale is not None else g_dev
    bits = _native.pack_relu_bits(pre.to(dev))
    wc = torch.cat([wl.t(), wr.t()], 1).to(dev)      # [Fi, 2 Fo]
    gin = torch.full((n, Fi), float('nan'), device=dev)
    rs = torch.rand(n, generator=g).to(dev) + 0.5
    gin_s = torch.full((n, Fi), float('nan'), device=dev)
    scratch = gin if Fi == Fo else torch.empty(n, Fo, device=dev)
    _native.sage_layer_forward(bwd.ptr, bwd.idx, gs, g_dev, wc, None, 'sum', False, scratch, gin,
                               hub=bwd.hub, save_agg=False, mask_bits=bits, row_scale=rs,
                               out_scaled=gin_s, variant=variant)
    # error scale: the sum of |terms| of the two products (hub rows add thousands of rows)
    ga = O.spmm(ei.flip(0), (gs.cpu()).abs(), n, 'sum').double()
    bound = ga @ wl.abs().double() + go.abs().double() @ wr.abs().double()
    ex = pre.double().requires_grad_(True)
    O.sage_conv(ex.relu(), ei, wl.double(), None, wr.double(), reduce).backward(go.double())
    assert_sum_close(gin, ref, ex.grad, abs_sum=bound + 1, what='fused input gradient')
    assert_close(gin_s, gin * rs.view(-1, 1), rtol=1e-6, atol=1e-6, what='row-scaled second output')
    assert bool((gin[pre.to(dev) <= 0] == 0).all())


def test_linear_dgrad_second_scaled_output(dev):
    """pygamd_linear_dgrad2: the row-scaled copy of the result comes out of the same pass, with
    the ReLU-bit epilogue applied to both and independently of `n_scaled`; interior and boundary
    tiles (M not a multiple of the tile)."""
    from pytorch_geometric_amd import _native
    g = gen(77)
    M, N, K = 70003, 96, 256
    go = torch.randn(M, N, generator=g).to(dev)
    w_t = (torch.randn(K, N, generator=g) * 0.1).to(dev)
    act = torch.randn(M, K, generator=g).to(dev)
    rs = (torch.rand(M, generator=g) + 0.5).to(dev)
    bits = _native.pack_relu_bits(act)
    plain = _native.linear_dgrad(go, w_t, relu_bits=bits)
    second = torch.full((M, K), float('nan'), device=dev)
    first = _native.linear_dgrad(go, w_t, row_scale=rs, relu_bits=bits, out_scaled=second)
    assert torch.equal(first, plain)
    assert torch.equal(second, plain * rs.view(-1, 1))
    second.fill_(float('nan'))
    first = _native.linear_dgrad(go, w_t, row_scale=rs, n_scaled=64, out_scaled=second)
    want = go @ w_t.t()
    want_first = want.clone()
    want_first[:, :64] *= rs.view(-1, 1)
    assert_close(first, want_first, rtol=1e-5, atol=1e-5, what='n_scaled columns')
    assert torch.equal(second, first * rs.view(-1, 1))


def test_linear_module_routes_large_inputs_to_the_own_gemm(dev):
    """nn.Linear: >= OWN_GEMM_MIN_ROWS float32 rows on the device run through LinearFunction
    (forward, input / weight / bias gradients on csrc/gemm.hip); values and gradients against the
    CPU F.linear, 3-D input included; small inputs stay on F.linear with identical semantics."""
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.nn import Linear
    g = gen(31)
    torch.manual_seed(4)
    lin = Linear(24, 10)
    calls = {'n': 0}
    real = _native.linear_forward

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    _native.linear_forward = counted
    try:
        for shape in ((Linear.OWN_GEMM_MIN_ROWS + 5, 24), (130, 130, 24), (50, 24)):
            x = torch.randn(*shape, generator=g)
            go = torch.randn(*shape[:-1], 10, generator=g)
            xr = x.clone().requires_grad_(True)
            ref = torch.nn.functional.linear(xr, lin.weight, lin.bias)
            lin.zero_grad()
            ref.backward(go)
            want = (ref.detach(), xr.grad, lin.weight.grad.clone(), lin.bias.grad.clone())
            dl = Linear(24, 10).to(dev)
            dl.load_state_dict(lin.state_dict())
            xd = x.to(dev).requires_grad_(True)
            before = calls['n']
            out = dl(xd)
            out.backward(go.to(dev))
            rows = x.numel() // 24
            # (the own GEMM is reached through the Python node or the C++ one, LinearAG)
            own = calls['n'] > before or 'LinearAG' in out.grad_fn.name()
            assert own == (rows >= Linear.OWN_GEMM_MIN_ROWS)
            assert_close(out, want[0], rtol=1e-5, atol=2e-5, what='linear out')
            assert_close(xd.grad, want[1], rtol=1e-5, atol=2e-5, what='linear grad x')
            scale = float(rows) ** 0.5  # weight / bias gradients sum over `rows` terms
            assert_close(dl.weight.grad, want[2], rtol=1e-5, atol=2e-5 * scale, what='grad W')
            assert_close(dl.bias.grad, want[3], rtol=1e-5, atol=2e-5 * scale, what='grad b')
    finally:
        _native.linear_forward = real


@pytest.fixture
def split_mode():
    from pytorch_geometric_amd import _native
    prev = _native.set_gemm_mode('split')
    yield
    _native.set_gemm_mode(prev)


@pytest.mark.parametrize('M,K,N', SHAPES)
def test_split_mode_forward_dgrad_wgrad(dev, split_mode, M, K, N):
    """PYGAMD_GEMM_SPLIT_BF16: the same entry points with every fp32 operand as three bf16 terms
    (six bf16 matrix products, fp32 accumulation).  Same acceptance as the exact mode: within 1e-5
    of the fp64 value, or at least as close to it as the CPU fp32 result (tests/_util.py)."""
    from pytorch_geometric_amd import _native
    assert _native.get_gemm_mode() == 'split'
    g = gen(M + K * 13 + N * 101 + 5)
    x, w, b, go = (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g),
                   torch.randn(N, generator=g), torch.randn(M, N, generator=g))
    out = _native.linear_forward(x.to(dev), w.to(dev), b.to(dev), relu=True)
    ex = (x.double() @ w.double().t() + b.double()).relu()
    bound = x.abs().double() @ w.abs().double().t() + b.abs().double()
    assert_sum_close(out, torch.nn.functional.linear(x, w, b).relu(), ex, abs_sum=bound,
                     what=f'split fwd {M}x{K}x{N}')
    out = _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev))
    assert_sum_close(out, go @ w, go.double() @ w.double(),
                     abs_sum=go.abs().double() @ w.abs().double(), what=f'split dgrad {M}x{K}x{N}')
    out, gb = _native.linear_wgrad(go.to(dev), x.to(dev), bias_grad=True)
    assert_sum_close(out, go.t() @ x, go.double().t() @ x.double(),
                     abs_sum=go.abs().double().t() @ x.abs().double(),
                     what=f'split wgrad {M}x{K}x{N}')
    assert_sum_close(gb, go.sum(0), go.double().sum(0), abs_sum=go.abs().double().sum(0),
                     what='split wgrad bias')


@pytest.mark.parametrize('M,K,N,K2', [
    (1, 1, 1, 0), (33, 7, 31, 0), (257, 512, 256, 0), (1000, 36, 130, 20), (4099, 100, 256, 100),
    (70001, 256, 47, 256), (129, 129, 47, 3), (65, 300, 130, 0), (9000, 64, 64, 0),
    (32, 8, 8, 0), (63, 8, 8, 0), (64, 8, 8, 0), (96, 8, 8, 8), (200000, 128, 128, 0),
    # from 32 k rows, 16-byte aligned operands: gemm_tn_skinny_kernel — narrow form (1..4 blocks of
    # g, one and two x tiles, ragged K, ragged last block, splits with 0 / 1 / 2 blocks behind the
    # pipeline) ...
    (70001, 256, 96, 0), (40003, 300, 128, 0), (33000, 64, 4, 0), (100003, 260, 48, 0),
    (32768, 256, 96, 0), (57345, 512, 36, 0),
    # wide g from 32 k rows (the tiled kernel; a streamed wide form was measured and dropped)
    (70001, 256, 256, 256), (40003, 100, 256, 100), (33000, 64, 200, 0), (50000, 300, 132, 20),
    (65536, 128, 256, 0),
])
def test_split_wgrad_production_schedule_against_the_in_register_one(dev, split_mode, M, K, N,
                                                                     K2):
    """The production split weight gradient converts every operand element once on its way into
    LDS and runs two wave groups per workgroup in anti-phase (even / odd blocks of a split, summed
    at the end); the lab switch (include/pyg_amd_lab.h) selects round 3's schedule, which converts
    next to the matrix instructions.  Same six-term products, different summation trees: both must
    meet the fp64 criterion of the other GEMM tests, agree with each other to rounding, and be
    deterministic — ragged tiles, two operands, unaligned rows (element-wise loads), odd block
    counts and many row splits included."""
    from pytorch_geometric_amd import _native
    g = gen(M + 3 * K + 7 * N)
    go = torch.randn(M, N, generator=g).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    x2 = torch.randn(M, K2, generator=g).to(dev) if K2 else None
    cases = [(go, x, x2)]
    if K > 1 and N > 1:  # rows that are not 16-byte aligned: the element-wise loads
        cases.append((torch.randn(M, N + 1, generator=g).to(dev)[:, 1:],
                      torch.randn(M, K + 1, generator=g).to(dev)[:, 1:], x2))
        cases.append((torch.randn(M, N + 1, generator=g).to(dev)[:, 1:], x, x2))
    for go_, x_, x2_ in cases:
        new, nb = _native.linear_wgrad(go_, x_, bias_grad=True, x2=x2_)
        _native.lab_set_wgrad_variant(1)
        try:
            old, ob = _native.linear_wgrad(go_, x_, bias_grad=True, x2=x2_)
        finally:
            _native.lab_set_wgrad_variant(0)
        cat = x_ if x2_ is None else torch.cat([x_, x2_], 1)
        exact = go_.double().t() @ cat.double()
        bound = go_.abs().double().t() @ cat.abs().double()
        assert_sum_close(new, old, exact, abs_sum=bound, what='weight gradient')
        assert_sum_close(nb, ob, go_.double().sum(0), abs_sum=go_.abs().double().sum(0),
                         what='bias gradient')
        again, ab = _native.linear_wgrad(go_, x_, bias_grad=True, x2=x2_)
        assert torch.equal(again, new) and torch.equal(ab, nb)


def test_split_mode_is_at_least_as_accurate_as_fp32(dev):
    """Error against fp64 of the two modes on one products-like shape (K = 512): the split mode's
    worst and mean errors must not exceed those of the exact fp32 instruction by more than 10 %."""
    from pytorch_geometric_amd import _native
    g = gen(123)
    x, w = torch.randn(4096, 512, generator=g), torch.randn(256, 512, generator=g)
    ex = x.double() @ w.double().t()
    scale = x.abs().double() @ w.abs().double().t()
    errs = {}
    for mode in ('fp32', 'split'):
        prev = _native.set_gemm_mode(mode)
        try:
            out = _native.linear_forward(x.to(dev), w.to(dev), None).cpu().double()
        finally:
            _native.set_gemm_mode(prev)
        rel = (out - ex).abs() / scale
        errs[mode] = (float(rel.max()), float(rel.mean()))
    assert errs['split'][0] <= 1.1 * errs['fp32'][0], errs
    assert errs['split'][1] <= 1.1 * errs['fp32'][1], errs


SMALL_M = [  # (M, K, N): sampled-batch blocks, the Cora shape, the FB15k-237 node count
    (1024, 512, 172), (1024, 512, 256), (2708, 1433, 16), (2708, 16, 7), (14541, 100, 32),
    (14541, 256, 256), (15360, 256, 512), (16384, 512, 256), (4000, 200, 96), (1500, 2048, 40),
]


@pytest.mark.parametrize('mode', ['fp32', 'split'])
@pytest.mark.parametrize('M,K,N', SMALL_M)
def test_small_m_gemm_forward_dgrad_wgrad(dev, mode, M, K, N):
    """1 k <= M < 16 k rows (VERDICT r3 missing #1): too few 128-row tiles to fill the chip, so the
    NT kernel runs 64 x 64 tiles and slices the reduction over grid.y (partial tiles combined in
    slice order by gemm_nt_splitk_epilogue, every epilogue applied there), the TN kernel splits M
    in 256-row pieces.  Same acceptance as the large shapes, in both arithmetic modes; bias + ReLU,
    row scales, ReLU-bit masks, the second scaled output and accumulate go through the combine
    pass."""
    import ctypes
    from pytorch_geometric_amd import _lib, _native
    prev = _native.set_gemm_mode(mode)
    try:
        g = gen(M + K * 13 + N * 101 + 9)
        x, w, b, go = (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g),
                       torch.randn(N, generator=g), torch.randn(M, N, generator=g))
        nb = ctypes.c_size_t(0)
        assert _lib.load().pygamd_linear_nt_workspace_bytes(M, N, K, ctypes.byref(nb)) == 0
        if K >= 128 and -(-M // 64) * -(-N // 64) < 128:
            assert nb.value > 0 and nb.value % (M * N * 4) == 0, 'expected a split over K'
        out = _native.linear_forward(x.to(dev), w.to(dev), b.to(dev), relu=True)
        ex = (x.double() @ w.double().t() + b.double()).relu()
        bound = x.abs().double() @ w.abs().double().t() + b.abs().double()
        assert_sum_close(out, torch.nn.functional.linear(x, w, b).relu(), ex, abs_sum=bound,
                         what=f'{mode} fwd {M}x{K}x{N}')
        # accumulate onto an existing output (the combine pass reads it)
        acc = torch.ones(M, N, device=dev)
        _native.linear_forward(x.to(dev), w.to(dev), None, out=acc, accumulate=True)
        assert_sum_close(acc - 1, x @ w.t(), x.double() @ w.double().t(), abs_sum=bound + 1,
                         what=f'{mode} fwd accumulate')
        # dgrad with every epilogue: 1/deg on the first columns, ReLU bits, second scaled output
        act = torch.randn(M, K, generator=g)
        rs = torch.rand(M, generator=g) + 0.5
        bits = _native.pack_relu_bits(act.to(dev))
        second = torch.full((M, K), float('nan'), device=dev)
        ns = (K // 2) // 4 * 4
        first = _native.linear_dgrad(go.to(dev), w.t().contiguous().to(dev), row_scale=rs.to(dev),
                                     n_scaled=ns, relu_bits=bits, out_scaled=second)
        want64 = go.double() @ w.double()
        want64[:, :ns] *= rs.double().view(-1, 1)
        want64 = want64 * (act > 0)
        want32 = (go @ w)
        want32[:, :ns] *= rs.view(-1, 1)
        want32 = want32 * (act > 0)
        bd = (go.abs().double() @ w.abs().double()) * rs.double().view(-1, 1).clamp(min=1)
        assert_sum_close(first, want32, want64, abs_sum=bd, what=f'{mode} dgrad {M}x{K}x{N}')
        assert torch.equal(second, first * rs.to(dev).view(-1, 1))
        gw, gb = _native.linear_wgrad(go.to(dev), x.to(dev), bias_grad=True)
        assert_sum_close(gw, go.t() @ x, go.double().t() @ x.double(),
                         abs_sum=go.abs().double().t() @ x.abs().double(),
                         what=f'{mode} wgrad {M}x{K}x{N}')
        assert_sum_close(gb, go.sum(0), go.double().sum(0), abs_sum=go.abs().double().sum(0),
                         what=f'{mode} wgrad bias')
    finally:
        _native.set_gemm_mode(prev)


def test_small_m_gemm_is_deterministic_and_unsliced_without_workspace(dev):
    """Two runs of a sliced launch agree bit for bit (slices are summed in order, no atomics); the
    raw C entry point without a workspace runs the same product unsliced."""
    import ctypes
    from pytorch_geometric_amd import _lib, _native
    lib = _lib.load()
    g = gen(404)
    M, K, N = 1024, 512, 172
    x, w = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
    a, b = _native.linear_forward(x, w, None), _native.linear_forward(x, w, None)
    assert torch.equal(a, b)
    out = torch.empty(M, N, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    assert lib.pygamd_linear_forward(x.data_ptr(), K, w.data_ptr(), K, None, M, K, N, 0, 0,
                                     out.data_ptr(), N, None, 0, st) == 0
    torch.cuda.synchronize()
    assert_close(out, a, rtol=1e-5, atol=1e-4, what='unsliced vs sliced')


SEGMM_SHAPES = [  # (K, N, blocks): K <= 128 and K % 4 == 0 take the convert-once split kernel
    (4, 1, 1), (8, 5, 1), (28, 31, 1), (32, 32, 1), (36, 33, 2), (64, 64, 1), (96, 65, 1),
    (100, 100, 5), (128, 128, 1), (128, 200, 1), (20, 130, 3), (100, 7, 1),
]


@pytest.mark.parametrize('K,N,blocks', SEGMM_SHAPES)
def test_segment_matmul_split_kernel(dev, K, N, blocks):
    """`segmm_split_kernel` (csrc/segmm.hip: weights converted once per workgroup into bf16 term
    planes, rows staged per wave): ragged segments incl. empty, 1-row, 31/32/33 and > 128-row ones,
    every K tail (chunks of 32, steps of 16, k-groups of 8) and column tail (halves of 64, blocks
    of 32), block-diagonal groups.  Values against fp64 within the |x|.|w| bound and no further
    from fp64 than the exact-instruction kernel on the same inputs (the split's acceptance rule);
    the transposed use (the input gradient) and garbage behind the operands' ends (padding must be
    staged as zeros, never multiplied as Inf * 0)."""
    from pytorch_geometric_amd import _native
    g = gen(K * 131 + N * 7 + blocks)
    lens = [0, 1, 31, 32, 33, 0, 127, 128, 129, 300, 5]
    ptr = [0]
    for n in lens:
        ptr.append(ptr[-1] + n)
    S, R = ptr[-1], len(lens)
    # the operands sit inside larger buffers filled with NaN: a read past an end shows up
    xbuf = torch.full((S + 2, blocks * K + 8), float('nan'))
    x = torch.randn(S, blocks * K, generator=g)
    xbuf[1:S + 1, 4:4 + blocks * K] = x
    w = torch.randn(R * blocks, K, N, generator=g) / K ** 0.5
    ex = torch.empty(S, blocks * N, dtype=torch.float64)
    bound = torch.empty_like(ex)
    for r in range(R):
        for b in range(blocks):
            xs = x[ptr[r]:ptr[r + 1], b * K:(b + 1) * K].double()
            ws = w[r * blocks + b].double()
            ex[ptr[r]:ptr[r + 1], b * N:(b + 1) * N] = xs @ ws
            bound[ptr[r]:ptr[r + 1], b * N:(b + 1) * N] = xs.abs() @ ws.abs()
    plan = _native.segmm_plan(tuple(ptr), dev, blocks)
    xg = xbuf.to(dev)[1:S + 1, 4:4 + blocks * K]
    assert xg.data_ptr() % 16 == 0 and xg.stride(0) % 4 == 0
    wg = w.to(dev)
    outs = {}
    for mode in ('fp32', 'split'):
        prev = _native.set_gemm_mode(mode)
        try:
            outs[mode] = _native.segment_matmul(xg, wg, plan, blocks=blocks).cpu().double()
        finally:
            _native.set_gemm_mode(prev)
    assert bool(torch.isfinite(outs['split']).all())
    tol = 1e-5 * bound + 1e-30
    err = {m: (o - ex).abs() for m, o in outs.items()}
    assert bool((err['split'] <= torch.maximum(tol, 2 * err['fp32'])).all()), \
        (float((err['split'] / (bound + 1e-30)).max()), float((err['fp32'] / (bound + 1e-30)).max()))
    rel = {m: float((e / (bound + 1e-30)).mean()) for m, e in err.items()}
    assert rel['split'] <= 1.25 * rel['fp32'] + 1e-9, rel
    # the transposed use: grad_x = grad_out @ W^T (N takes the role of K; only when it fits)
    go = torch.randn(S, blocks * N, generator=g)
    prev = _native.set_gemm_mode('split')
    try:
        gx = _native.segment_matmul(go.to(dev), wg, plan, transpose_w=True, blocks=blocks)
    finally:
        _native.set_gemm_mode(prev)
    exg = torch.empty(S, blocks * K, dtype=torch.float64)
    bg = torch.empty_like(exg)
    for r in range(R):
        for b in range(blocks):
            gs = go[ptr[r]:ptr[r + 1], b * N:(b + 1) * N].double()
            wt = w[r * blocks + b].double().t()
            exg[ptr[r]:ptr[r + 1], b * K:(b + 1) * K] = gs @ wt
            bg[ptr[r]:ptr[r + 1], b * K:(b + 1) * K] = gs.abs() @ wt.abs()
    assert bool(((gx.cpu().double() - exg).abs() <= 1e-5 * bg + 1e-30).all())


@pytest.mark.parametrize('mode', ['split', 'fp32'])
@pytest.mark.parametrize('K,N,blocks', [(100, 100, 5), (36, 20, 2), (130, 33, 1)])
def test_segment_matmul_row_index_operands(dev, mode, K, N, blocks):
    """`x_rows` / `g_rows` of pygamd_segment_matmul(_wgrad): operand row s is `x[x_rows[s]]` —
    bit-equal to gathering the rows first and running the plain call (same kernels, same order),
    for the convert-once split kernel, the fp32 kernel (K = 130) and the weight gradient; and the
    autograd node RGCNConv uses (`segment_matmul_sum`: grouped GEMM + per-destination sum, whose
    backward reads the destinations' gradient rows through the index) against plain PyTorch."""
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.edge_index import EdgeIndex
    from pytorch_geometric_amd.utils._segment_matmul import segment_matmul_sum
    g = gen(K + 3 * N + blocks)
    lens = [0, 3, 129, 64, 1, 200]
    ptr = [0]
    for n in lens:
        ptr.append(ptr[-1] + n)
    S, R, n_small = ptr[-1], len(lens), 57
    small = torch.randn(n_small, blocks * K, generator=g).to(dev)
    rows = torch.randint(0, n_small, (S, ), generator=g).to(dev)
    w = (torch.randn(R * blocks, K, N, generator=g) / K ** 0.5).to(dev)
    plan = _native.segmm_plan(tuple(ptr), dev, blocks)
    prev = _native.set_gemm_mode(mode)
    try:
        got = _native.segment_matmul(small, w, plan, blocks=blocks, x_rows=rows)
        want = _native.segment_matmul(small[rows], w, plan, blocks=blocks)
        assert torch.equal(got, want)
        gsmall = torch.randn(n_small, blocks * N, generator=g).to(dev)
        xs = torch.randn(S, blocks * K, generator=g).to(dev)
        gw = _native.segment_matmul_wgrad(xs, gsmall, plan, R * blocks, blocks, g_rows=rows)
        gw_want = _native.segment_matmul_wgrad(xs, gsmall[rows], plan, R * blocks, blocks)
        assert_close(gw, gw_want, rtol=1e-5, atol=1e-5, what='wgrad through the row index')
        # the autograd node: rows of the product summed per destination
        n_dst = 40
        dst = torch.randint(0, n_dst, (S, ), generator=g).to(dev)
        out_graph = EdgeIndex(torch.stack([torch.arange(S, device=dev), dst]), (S, n_dst),
                              sort_order='row', validate=False)
        w4 = w.view(R, blocks, K, N) if blocks > 1 else w
        xin = xs.clone().requires_grad_(True)
        wp = w4.clone().requires_grad_(True)
        out = segment_matmul_sum(xin, tuple(ptr), wp, out_graph)
        go = torch.randn(n_dst, blocks * N, generator=g).to(dev)
        out.backward(go)
        xr = xs.double().cpu().requires_grad_(True)
        wr = w.double().cpu().requires_grad_(True)
        t = torch.cat([torch.cat([xr[ptr[r]:ptr[r + 1], b * K:(b + 1) * K] @ wr[r * blocks + b]
                                  for b in range(blocks)], dim=1) for r in range(R)])
        ref = torch.zeros(n_dst, blocks * N, dtype=torch.float64).index_add_(0, dst.cpu(), t)
        ref.backward(go.double().cpu())
        assert_close(out, ref.detach().float(), rtol=1e-5, atol=5e-5, what='segment_matmul_sum')
        assert_close(xin.grad, xr.grad.float(), rtol=1e-5, atol=5e-5, what='grad inputs')
        assert_close(wp.grad.reshape(wr.shape), wr.grad.float(), rtol=1e-4, atol=1e-4,
                     what='grad weights')
    finally:
        _native.set_gemm_mode(prev)


def test_split_nonfinite_matches_reference(dev):
    """Non-finite operands.  The reference's `F.linear` gives `w * Inf = +-Inf`
    (and NaN for `0 * Inf`, `Inf - Inf`, NaN operands).  The default arithmetic — every fp32 operand
    as x1 + x2 + x3 in bf16 — cannot represent Inf that way (`Inf - bf16(Inf)` is NaN), so an
    output that DEPENDS on a non-finite operand comes out NaN where the reference says +-Inf.
    What is guaranteed, and checked here against `F.linear` on the CPU:
      * outputs that do not depend on a non-finite operand are unaffected (same values as a clean
        run) — a poisoned row / column never spreads;
      * every output the reference makes non-finite is non-finite here too (NaN or the same Inf):
        `isfinite` masks agree exactly, so `torch.isfinite(out).all()` guards behave the same;
      * `set_gemm_mode('fp32')` (the exact `v_mfma_f32_32x32x2_f32` kernels) reproduces the
        reference's class element by element: +-Inf where it says +-Inf, NaN where it says NaN."""
    from pytorch_geometric_amd import _native
    g = gen(77)
    M, K, N = 3000, 96, 40
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    go = torch.randn(M, N, generator=g)
    clean = torch.nn.functional.linear(x, w, b)
    xb = x.clone()
    xb[5, 3] = float('inf')
    xb[17, 0] = float('-inf')
    xb[40, 9] = float('nan')
    xb[99, 1], xb[99, 2] = float('inf'), float('-inf')       # Inf - Inf in one row
    wb = w.clone()
    wb[7, 11] = float('inf')
    wb[8, 3] = 0.0                                            # 0 * Inf for row 5
    ref = torch.nn.functional.linear(xb, wb, b)
    ref_dg = go @ wb
    ref_wg = go.t() @ xb
    for mode in ('split', 'fp32'):
        prev = _native.set_gemm_mode(mode)
        try:
            out = _native.linear_forward(xb.to(dev), wb.to(dev), b.to(dev)).cpu()
            dg = _native.linear_dgrad(go.to(dev), wb.t().contiguous().to(dev)).cpu()
            wg = _native.linear_wgrad(go.to(dev), xb.to(dev)).cpu()
        finally:
            _native.set_gemm_mode(prev)
        for got, want, what in ((out, ref, 'forward'), (dg, ref_dg, 'dgrad'), (wg, ref_wg, 'wgrad')):
            fin = torch.isfinite(want)
            assert torch.equal(torch.isfinite(got), fin), f'{mode} {what}: finite masks differ'
            assert_close(got[fin], want[fin], rtol=1e-5, atol=1e-4, what=f'{mode} {what} finite part')
            if mode == 'fp32':   # the reference's class exactly
                assert torch.equal(torch.isnan(got), torch.isnan(want)), f'{what}: NaN class'
                inf = torch.isinf(want)
                assert torch.equal(got[inf], want[inf]), f'{what}: signed infinities'
        if mode == 'split':     # clean rows / columns are bit-identical to a clean run
            ok_rows = torch.ones(M, dtype=torch.bool)
            ok_rows[[5, 17, 40, 99]] = False
            ok_cols = torch.ones(N, dtype=torch.bool)
            ok_cols[7] = False
            prev = _native.set_gemm_mode('split')
            try:
                base = _native.linear_forward(x.to(dev), w.to(dev), b.to(dev)).cpu()
            finally:
                _native.set_gemm_mode(prev)
            # (column 8 of the clean run differs by the zeroed weight: compare the rest)
            ok_cols8 = ok_cols.clone()
            ok_cols8[8] = False
            assert torch.equal(out[ok_rows][:, ok_cols8], base[ok_rows][:, ok_cols8])
            assert_close(base, clean, rtol=1e-5, atol=1e-4, what='clean split run')

"""Host logic of the fused GraphSAGE stack (nn/models/_fused_sage.py) WITHOUT a GPU: every native
entry point the stack calls is replaced — inside this test only — by a plain-torch stand-in that
honours the same arguments (strided halves of `[agg | x]` buffers, `accumulate`, the `1/deg` row
scale of the dgrad, the ReLU mask as floats or as 32 x 32-bit tiles, the bias gradient returned by
the weight gradient, the two-operand weight gradient).  What is checked is the WIRING: which
buffers, masks and operands each step of the forward / backward schedule hands to which kernel.
Values and gradients must equal the oracle's layer-by-layer GraphSAGE under autograd.  The kernels
themselves are tested on the device (tests/test_gpu_*.py); nothing here ships."""
import pytest
import torch

from oracle import pyg_oracle as O
from tests._util import assert_close, gen, random_graph


class _Csr:
    def __init__(self, key, other, n_rows, n_cols):
        order = torch.sort(key, stable=True).indices
        self.ptr = torch._convert_indices_from_coo_to_csr(key[order], n_rows)
        self.idx, self.perm = other[order].contiguous(), order
        self.n_rows, self.n_cols, self.hub = n_rows, n_cols, None

    def inv_degree(self):
        return 1.0 / (self.ptr[1:] - self.ptr[:-1]).clamp(min=1).to(torch.float32)


class _Graph:
    """Duck type of pytorch_geometric_amd.EdgeIndex for FusedSageStack."""

    def __init__(self, ei, n):
        self._fwd, self._bwd = _Csr(ei[1], ei[0], n, n), _Csr(ei[0], ei[1], n, n)
        self.edge_index, self.num_edges = ei, ei.size(1)

    def by_dst(self):
        return self._fwd

    def by_src(self):
        return self._bwd


def _aggregate(ptr, idx, x, reduce):
    n = ptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n), ptr[1:] - ptr[:-1])
    idx = idx[int(ptr[0]):int(ptr[-1])]  # (a clipped pointer covers a prefix of the slots)
    out = torch.zeros(n, x.size(1)).index_add_(0, rows, x[idx])
    if reduce == 'mean':
        out = out / (ptr[1:] - ptr[:-1]).clamp(min=1).view(-1, 1)
    return out


def _unpack_bits(bits, n, f):
    w = bits.to(torch.int64) & 0xffffffff                     # [tile, block, row in tile]
    shifts = torch.arange(32)
    m = ((w.unsqueeze(-1) >> shifts) & 1).permute(0, 2, 1, 3)  # [tile, row, block, bit]
    return m.reshape(m.size(0) * 32, -1)[:n, :f].bool()


@pytest.fixture
def fake_native(monkeypatch):
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.nn.models import _fused_sage
    log = []
    zstore = {}

    def spmm_csr(ptr, idx, x, reduce, *, n_rows=None, hub=None, out=None, accumulate=False,
                 src_scale=None, relu_mask=None, relu_bits=None, src_bits=None,
                 src_bits_set=None, **kw):
        assert not kw, kw
        xs = x if src_scale is None else x * src_scale.view(-1, 1)
        if src_bits is not None:  # only the rows whose bit is set are read
            assert src_scale is None and reduce in ('sum', 'mean')
            live = torch.tensor([(int(src_bits[i >> 5]) >> (i & 31)) & 1
                                 for i in range(x.size(0))], dtype=torch.bool)
            assert int(src_bits_set) == int(live.sum())
            assert bool((xs[~live] == 0).all()), 'a row with a clear bit is not all-zero'
        res = _aggregate(ptr, idx, xs, reduce)
        if accumulate:
            res = res + out
        assert relu_mask is None or relu_bits is None
        if relu_mask is not None:
            res = torch.where(relu_mask > 0, res, torch.zeros_like(res))
        if relu_bits is not None:
            res = torch.where(_unpack_bits(relu_bits, *res.shape), res, torch.zeros_like(res))
        log.append(('spmm', reduce, accumulate, relu_mask is not None, relu_bits is not None)
                   + (('src_bits', ) if src_bits is not None else ()))
        if out is None:
            return res
        out.copy_(res)
        return out

    def sage_layer_forward(ptr, idx, x_gather, x_root, w, bias, reduce, relu, agg, out, hub=None,
                           save_agg=True, relu_bits=None, mask_bits=None, row_scale=None,
                           out_scaled=None, gather_width=None, compressed_out=None, **kw):
        assert not kw, kw
        if gather_width is not None:  # the previous layer's compressed copy of its output
            assert x_gather.dtype == torch.int32
            x_gather = zstore[x_gather.data_ptr()]
            assert x_gather.size(1) == gather_width and torch.equal(x_gather, x_root)
        a = _aggregate(ptr, idx, x_gather, reduce)
        y = torch.cat([a, x_root], 1) @ w.t()
        if bias is not None:
            y = y + bias
        if relu:
            y = y.relu()
        if mask_bits is not None:  # the launch is a layer's input gradient
            y = torch.where(_unpack_bits(mask_bits, *y.shape), y, torch.zeros_like(y))
        if save_agg:
            agg.copy_(a)
        if relu_bits is not None:
            assert relu
            relu_bits.copy_(_native.pack_relu_bits(y))
        out.copy_(y)
        if compressed_out is not None:
            assert relu and compressed_out.dtype == torch.int32
            assert compressed_out.shape == (y.size(0), _native.compressed_pitch(y.size(1)))
            zstore[compressed_out.data_ptr()] = y.clone()
        if out_scaled is not None:
            out_scaled.copy_(y * row_scale.view(-1, 1))
        # (root rows read from a dense tensor = the layer input itself, not a half of [agg | x])
        log.append(('fused_layer', x_root.is_contiguous(), relu_bits is not None)
                   + (('z_in', ) if gather_width is not None else ())
                   + (('z_out', ) if compressed_out is not None else ())
                   if not (mask_bits is not None or not save_agg) else
                   ('fused_layer_bwd', mask_bits is not None, out_scaled is not None))
        return out

    def rows_pack(g, row_scale=None, *, scaled=None, copy=None, count=True):
        live = (g != 0).any(dim=1) | g.isnan().any(dim=1)
        words = torch.zeros((g.size(0) + 31) // 32, dtype=torch.int64)
        for i in torch.nonzero(live).flatten().tolist():
            words[i >> 5] |= 1 << (i & 31)
        if scaled is not None:
            scaled.zero_()
            scaled[:, :g.size(1)] = g if row_scale is None else g * row_scale.view(-1, 1)
        if copy is not None:
            copy.zero_()
            copy[:, :g.size(1)] = g
        log.append(('rows_pack', scaled is not None, copy is not None))
        return words, (live.sum().view(1) if count else None)

    def linear_forward(x, w, bias=None, relu=False, out=None, accumulate=False):
        y = x @ w.t()
        if bias is not None:
            y = y + bias
        if relu:
            y = y.relu()
        if out is None:
            return y
        out.copy_(y + out if accumulate else y)
        return out

    def linear_dgrad(g, w_t, row_scale=None, n_scaled=0, out=None, accumulate=False,
                     relu_mask=None, relu_bits=None, out_scaled=None):
        y = g @ w_t.t()
        if row_scale is not None and n_scaled:
            y[:, :n_scaled] *= row_scale.view(-1, 1)
        assert relu_mask is None or relu_bits is None
        if relu_mask is not None:
            y = torch.where(relu_mask > 0, y, torch.zeros_like(y))
        if relu_bits is not None:
            y = torch.where(_unpack_bits(relu_bits, *y.shape), y, torch.zeros_like(y))
        if out_scaled is not None:
            out_scaled.copy_(y * row_scale.view(-1, 1))
        log.append(('dgrad', relu_mask is not None, relu_bits is not None, out_scaled is not None))
        return y

    def linear_wgrad(g, x, out=None, accumulate=False, wgs_per_cu=0, bias_grad=False, x2=None):
        both = x if x2 is None else torch.cat([x, x2], 1)
        gw = g.t() @ both
        log.append(('wgrad', bias_grad, x2 is not None))
        return (gw, g.sum(0)) if bias_grad else gw

    for name, fn in dict(spmm_csr=spmm_csr, sage_layer_forward=sage_layer_forward,
                         rows_pack=rows_pack,
                         linear_forward=linear_forward, linear_dgrad=linear_dgrad,
                         linear_wgrad=linear_wgrad).items():
        monkeypatch.setattr(_native, name, fn)
    monkeypatch.setattr(_native, 'sage_layer_forward_supported', lambda F, Fo, r: F % 4 == 0)
    monkeypatch.setattr(_native, 'colsum', lambda g: g.sum(0))
    monkeypatch.setattr(_native, 'relu_backward_colsum',
                        lambda g, h, want: (torch.where(h > 0, g, torch.zeros_like(g)),
                                            torch.where(h > 0, g, torch.zeros_like(g)).sum(0)
                                            if want else None))
    monkeypatch.setattr(_fused_sage, 'GEMM_BACKEND', 'own')
    monkeypatch.setattr(_fused_sage, 'FUSE_LAYER', True)
    monkeypatch.setattr(_fused_sage, 'OVERLAP_WGRAD', False)
    return log


@pytest.mark.parametrize('dims,aggr,x_grad', [
    ((100, 256, 256, 47), 'mean', False),   # the headline shape: post, post, pre
    ((100, 256, 256, 47), 'mean', True),
    ((16, 8, 24, 12), 'sum', True),          # pre, post, pre
    ((10, 20, 6), 'mean', True),             # F % 4 != 0: the [agg | x] buffer path of layer 1
    ((32, 32), 'sum', False),                # a single layer
])
@pytest.mark.parametrize('bias', [True, False])
@pytest.mark.parametrize('fuse_bwd', [True, False])
def test_fused_stack_wiring_against_the_oracle(fake_native, monkeypatch, dims, aggr, x_grad, bias,
                                               fuse_bwd):
    from pytorch_geometric_amd.nn.models import _fused_sage
    from pytorch_geometric_amd.nn.models._fused_sage import FusedSageStack
    monkeypatch.setattr(_fused_sage, 'FUSE_BWD', fuse_bwd)
    monkeypatch.setattr(_fused_sage, 'COMPRESS_ROWS', True)  # (opt-in: the wiring is checked)
    n = 75  # three 32-row bit tiles, the last one partial
    g = gen(sum(dims) + n)
    ei = random_graph(n, n, 600, seed=dims[0], skew=True)
    ei[1][ei[1] == 7] = 8  # a node without in-edges
    x = torch.randn(n, dims[0], generator=g)
    params = []
    for fi, fo in zip(dims[:-1], dims[1:]):
        params.append((torch.randn(fo, fi, generator=g) * 0.3,
                       torch.randn(fo, generator=g) if bias else None,
                       torch.randn(fo, fi, generator=g) * 0.3))
    go = torch.randn(n, dims[-1], generator=g)
    if dims[-1] == 47:  # a loss on a training split: most rows of the incoming gradient are zero
        go[torch.rand(n, generator=g) < 0.85] = 0

    def leaves():
        xs = x.clone().requires_grad_(x_grad)
        ps = [tuple(None if t is None else t.clone().requires_grad_(True) for t in p)
              for p in params]
        return xs, ps

    xr, pr = leaves()
    ref = O.graphsage(xr, ei, pr, aggr)
    ref.backward(go)
    xf, pf = leaves()
    flat = [t for p in pf for t in p]
    out = FusedSageStack.apply(xf, _Graph(ei, n), aggr, True, *flat)
    out.backward(go)
    assert_close(out, ref, rtol=1e-4, atol=1e-4, what='fused stack output')
    for layer, (a, b) in enumerate(zip(pf, pr)):
        for name, t, r in zip(('W_l', 'b', 'W_r'), a, b):
            if t is not None:
                assert_close(t.grad, r.grad, rtol=1e-4, atol=2e-4, what=f'layer {layer} {name}')
    if x_grad:
        assert_close(xf.grad, xr.grad, rtol=1e-4, atol=2e-4, what='grad x')
    else:
        assert xf.grad is None

    # the schedule itself: no stand-alone ReLU / bias pass, bits wherever a fused layer made them
    kinds = [e[0] for e in fake_native]
    L = len(dims) - 1
    assert kinds.count('wgrad') == L
    for e in fake_native:
        if e[0] == 'wgrad':
            assert e[1] == bias          # the bias gradient rides on the weight gradient
    if dims == (100, 256, 256, 47):
        # the output layer's backward lays its gradient out in one pass that also finds the live
        # rows, and its transposed aggregation takes their bitmap
        assert [e for e in fake_native if e[0] == 'rows_pack'] == [('rows_pack', True, True)]
        assert [e for e in fake_native if e[0] == 'spmm' and e[-1] == 'src_bits'] == [
            ('spmm', 'sum', False, False, False, 'src_bits')]
        fused = [e for e in fake_native if e[0] == 'fused_layer']
        assert [e[1] for e in fused] == [True, False]   # layer 1 roots on x itself (no copy)
        # layer 1 writes its ReLU output once more as compressed rows; layer 2 gathers those
        assert fused[0][3:] == ('z_out', ) and fused[1][3:] == ('z_in', )
        assert ('wgrad', bias, True) in fake_native      # ... and its wgrad takes [agg | x] apart
        # ReLU backward of h1 and h2 in the epilogues of the kernels producing their gradients,
        # and never from the float activations: both came out of the one-kernel layer forward
        floats = [e for e in fake_native if (e[0] == 'spmm' and e[3]) or (e[0] == 'dgrad' and e[1])]
        assert not floats
        bwd = [e for e in fake_native if e[0] == 'fused_layer_bwd']
        if fuse_bwd:
            # layer 2's input gradient = ONE launch (no dgrad GEMM, no transposed SpMM at width
            # 256); the dgrad of layer 3 hands it the 1/deg-scaled rows as a second output
            # (with a gradient for x, layer 1 runs the same way — unmasked — and layer 2 hands it
            # its scaled rows)
            assert bwd == ([('fused_layer_bwd', True, x_grad)]
                           + ([('fused_layer_bwd', False, False)] if x_grad else []))
            assert [e for e in fake_native if e[0] == 'dgrad'] == [('dgrad', False, True, True)]
            assert not [e for e in fake_native if e[0] == 'spmm' and e[4]]
        else:
            assert not bwd
            masked = [e for e in fake_native
                      if (e[0] == 'spmm' and e[4]) or (e[0] == 'dgrad' and e[2])]
            assert len(masked) == 2                      # h1 (transposed SpMM) and h2 (dgrad)
    if dims == (16, 8, 24, 12):
        # pre, post, pre: the post layer's input came out of a GEMM + SpMM pair (no bit mask), so
        # its gradient keeps the two-launch form with the float mask
        assert not [e for e in fake_native if e[0] == 'fused_layer_bwd']


@pytest.mark.parametrize('dims', [(12, 20, 20, 5), (16, 8, 12)])
def test_library_gemm_schedule_wiring(fake_native, monkeypatch, dims):
    """PYGAMD_GEMM=lib: the dense transforms go to torch.mm / addmm, the transposed SpMM takes the
    `1/deg` as a per-source scale, ReLU backward + bias gradient stay one stand-alone pass."""
    from pytorch_geometric_amd.nn.models import _fused_sage
    from pytorch_geometric_amd.nn.models._fused_sage import FusedSageStack
    monkeypatch.setattr(_fused_sage, 'GEMM_BACKEND', 'lib')
    n = 40
    g = gen(sum(dims))
    ei = random_graph(n, n, 300, seed=3)
    x = torch.randn(n, dims[0], generator=g)
    params = [(torch.randn(fo, fi, generator=g) * 0.3, torch.randn(fo, generator=g),
               torch.randn(fo, fi, generator=g) * 0.3) for fi, fo in zip(dims[:-1], dims[1:])]
    go = torch.randn(n, dims[-1], generator=g)
    res = []
    for fused in (False, True):
        xs = x.clone().requires_grad_(True)
        ps = [tuple(t.clone().requires_grad_(True) for t in p) for p in params]
        if fused:
            out = FusedSageStack.apply(xs, _Graph(ei, n), 'mean', True, *[t for p in ps for t in p])
        else:
            out = O.graphsage(xs, ei, ps, 'mean')
        out.backward(go)
        res.append([out.detach(), xs.grad] + [t.grad for p in ps for t in p])
    for a, b in zip(*res):
        assert_close(b, a, rtol=1e-4, atol=2e-4, what='library-GEMM schedule')
    assert not [e for e in fake_native if e[0] in ('wgrad', 'dgrad', 'fused_layer')]


def _sampled_batch(nodes_per_hop, fanout, g):
    """A NeighborLoader-shaped batch: nodes listed hop by hop, the edges of hop h lead from nodes of
    hops <= h + 1 to the nodes of hop h, destination-sorted (so the whole list is, too)."""
    starts = [0]
    for c in nodes_per_hop:
        starts.append(starts[-1] + c)
    src, dst, per_hop = [], [], []
    for h in range(len(nodes_per_hop) - 1):
        count = 0
        for d in range(starts[h], starts[h + 1]):
            k = int(torch.randint(0, fanout + 1, (1, ), generator=g))
            src.append(torch.randint(0, starts[h + 2], (k, ), generator=g))
            dst.append(torch.full((k, ), d))
            count += k
        per_hop.append(count)
    return torch.stack([torch.cat(src), torch.cat(dst)]), per_hop


@pytest.mark.parametrize('own', [False, True])
@pytest.mark.parametrize('aggr', ['mean', 'sum'])
def test_hop_aware_stack_wiring(fake_native, monkeypatch, own, aggr):
    """nn/models/_fused_sage_hops.py: every layer computes only the rows the next one consumes,
    from a prefix of the destination-sorted edge list.  The seed rows of its output, and the
    gradients of a loss on them, must equal the full GraphSAGE on the sampled subgraph (rows
    further out see fewer neighbours by construction: trim_to_layer,
    utils/_trim_to_layer.py:44-127)."""
    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.nn.models import _fused_sage
    from pytorch_geometric_amd.nn.models._fused_sage_hops import FusedSageHopStack

    def gather_scatter_add(x, gather_idx, scatter_idx, n_out, scale=None, w=None, out=None):
        assert w is None
        v = x[gather_idx] if scale is None else x[gather_idx] * scale[gather_idx].view(-1, 1)
        if out is None:
            out = torch.zeros(n_out, x.size(1))
        out.index_add_(0, scatter_idx, v)
        return out

    monkeypatch.setattr(_native, 'gather_scatter_add', gather_scatter_add)
    monkeypatch.setattr(_fused_sage, 'OWN_GEMM_MIN_ROWS', 0 if own else 1 << 30)
    g = gen(17)
    nodes_per_hop = [5, 9, 14, 20]
    ei, edges_per_hop = _sampled_batch(nodes_per_hop, 4, g)
    n, dims = sum(nodes_per_hop), (12, 16, 16, 7)
    x = torch.randn(n, dims[0], generator=g)
    params = [(torch.randn(fo, fi, generator=g) * 0.3, torch.randn(fo, generator=g),
               torch.randn(fo, fi, generator=g) * 0.3) for fi, fo in zip(dims[:-1], dims[1:])]
    seeds = nodes_per_hop[0]
    go = torch.randn(seeds, dims[-1], generator=g)
    res = []
    for fused in (False, True):
        xs = x.clone().requires_grad_(True)
        ps = [tuple(t.clone().requires_grad_(True) for t in p) for p in params]
        if fused:
            out = FusedSageHopStack.apply(xs, _Graph(ei, n), aggr, nodes_per_hop, edges_per_hop,
                                          *[t for p in ps for t in p])
            assert out.size(0) == n - nodes_per_hop[-1] - nodes_per_hop[-2]
        else:
            out = O.graphsage(xs, ei, ps, aggr)
        out[:seeds].backward(go)
        res.append([out[:seeds].detach(), xs.grad] + [t.grad for p in ps for t in p])
    for a, b in zip(*res):
        assert_close(b, a, rtol=1e-4, atol=2e-4, what=f'hop-aware stack (own={own})')
    assert bool([e for e in fake_native if e[0] == 'wgrad']) == own


def test_fused_stack_steps_aside_for_lazy_or_foreign_parameters():
    """a lazily initialised model (in_channels = -1) has UninitializedParameter weights
    until its first ORIGINAL forward; the fused route must not be taken then (nor for parameters of
    another dtype / device than the input)."""
    from pytorch_geometric_amd.nn.models._fused_sage import params_ready

    class Lin:
        def __init__(self, weight, bias=None):
            self.weight, self.bias = weight, bias

    class Conv:
        def __init__(self, wl, wr, b=None):
            self.lin_l, self.lin_r = Lin(wl, b), Lin(wr)

    x = torch.randn(4, 3)
    w = torch.nn.Parameter(torch.randn(5, 3))
    assert params_ready(Conv(w, w, torch.nn.Parameter(torch.zeros(5))), x)
    lazy = torch.nn.parameter.UninitializedParameter()
    assert not params_ready(Conv(lazy, w), x)
    assert not params_ready(Conv(w, lazy), x)
    assert not params_ready(Conv(w.double(), w), x)
    assert not params_ready(Conv(w, w, torch.nn.Parameter(torch.zeros(5, dtype=torch.float64))), x)
    assert not params_ready(Conv(w.to('meta'), w), x)


@pytest.mark.parametrize('own', [False, True])
@pytest.mark.parametrize('aggr', ['mean', 'sum'])
def test_padded_hop_stack_wiring(fake_native, monkeypatch, own, aggr):
    """FusedSagePaddedHopStack (the static-shape batch a captured training step runs on): node ids
    are block positions with padding rows, each hop has its own CSR pointer over its destination
    block and its real-edge count on the device.  Seed rows, parameter gradients and the input
    gradient on the real rows must equal the full GraphSAGE on the compact sampled subgraph; padding
    rows hold garbage on the way in and get exactly zero gradient."""
    from types import SimpleNamespace

    from pytorch_geometric_amd import _native
    from pytorch_geometric_amd.nn.models import _fused_sage
    from pytorch_geometric_amd.nn.models._fused_sage_hops import FusedSagePaddedHopStack

    def gather_scatter_add(x, gather_idx, scatter_idx, n_out, scale=None, w=None, out=None,
                           n_valid=None):
        assert w is None and n_valid is not None
        n = int(n_valid)
        gi, si = gather_idx[:n], scatter_idx[:n]
        v = x[gi] if scale is None else x[gi] * scale[gi].view(-1, 1)
        out.index_add_(0, si, v)
        return out

    monkeypatch.setattr(_native, 'gather_scatter_add', gather_scatter_add)
    monkeypatch.setattr(_fused_sage, 'OWN_GEMM_MIN_ROWS', 0 if own else 1 << 30)
    g = gen(23)
    nodes_per_hop = [4, 7, 11, 16]                   # compact batch: seeds + 3 hops
    ei, edges_per_hop = _sampled_batch(nodes_per_hop, 3, g)
    L = 3
    n, dims = sum(nodes_per_hop), (8, 12, 12, 5)
    caps = [4, 9, 15, 21]                            # block capacities >= the real counts
    bases = [0]
    for c in caps:
        bases.append(bases[-1] + c)
    starts = [0]
    for c in nodes_per_hop:
        starts.append(starts[-1] + c)
    # compact id -> padded id
    pad_of = torch.empty(n, dtype=torch.long)
    for b in range(L + 1):
        pad_of[starts[b]:starts[b + 1]] = torch.arange(nodes_per_hop[b]) + bases[b]
    ptrs, rows, cols, n_edges = [], [], [], []
    e0 = 0
    for h in range(L):
        e1 = e0 + edges_per_hop[h]
        src, dst = ei[0, e0:e1], ei[1, e0:e1]
        deg = torch.bincount(dst - starts[h], minlength=caps[h])      # padding rows: degree 0
        ptrs.append(torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)]))
        cap_e = edges_per_hop[h] + 5                                   # zero-filled tail
        r = torch.zeros(cap_e, dtype=torch.long)
        c = torch.zeros(cap_e, dtype=torch.long)
        r[:e1 - e0], c[:e1 - e0] = pad_of[src], pad_of[dst]
        rows.append(r)
        cols.append(c)
        n_edges.append(torch.tensor([e1 - e0]))
        e0 = e1
    batch = SimpleNamespace(bases=bases, ptrs=ptrs, rows=rows, cols=cols, n_edges=n_edges)
    x = torch.randn(n, dims[0], generator=g)
    x_pad = torch.randn(bases[-1], dims[0], generator=g) * 50.0        # garbage in the padding
    x_pad[pad_of] = x
    params = [(torch.randn(fo, fi, generator=g) * 0.3, torch.randn(fo, generator=g),
               torch.randn(fo, fi, generator=g) * 0.3) for fi, fo in zip(dims[:-1], dims[1:])]
    seeds = nodes_per_hop[0]
    go = torch.randn(seeds, dims[-1], generator=g)
    # reference: the full model on the compact subgraph
    xs = x.clone().requires_grad_(True)
    ps = [tuple(t.clone().requires_grad_(True) for t in p) for p in params]
    ref = O.graphsage(xs, ei, ps, aggr)
    ref[:seeds].backward(go)
    want = [ref[:seeds].detach(), xs.grad] + [t.grad for p in ps for t in p]
    xp = x_pad.clone().requires_grad_(True)
    pp = [tuple(t.clone().requires_grad_(True) for t in p) for p in params]
    out = FusedSagePaddedHopStack.apply(xp, batch, aggr, *[t for p in pp for t in p])
    assert out.shape == (bases[1], dims[-1])
    out[:seeds].backward(go)
    got = [out[:seeds].detach(), xp.grad[pad_of]] + [t.grad for p in pp for t in p]
    for a, b in zip(want, got):
        assert_close(b, a, rtol=1e-4, atol=2e-4, what=f'padded hop stack (own={own})')
    padding = torch.ones(bases[-1], dtype=torch.bool)
    padding[pad_of] = False
    assert bool((xp.grad[padding] == 0).all())
    assert bool([e for e in fake_native if e[0] == 'wgrad']) == own

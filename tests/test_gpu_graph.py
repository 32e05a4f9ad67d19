"""Integer side: bit-exact parity with the oracle (= torch.sort(stable) /
_convert_indices_from_coo_to_csr) and the reference's golden vectors."""
import pytest
import torch

from oracle import pyg_oracle as O
from tests._util import assert_close, gen, random_graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('n,hi', [(0, 5), (1, 1), (500, 37), (100_000, 1000), (300_000, 3)])
def test_index_sort_bit_exact(dev, dtype, n, hi):
    import pytorch_geometric_amd as pga
    keys = torch.randint(0, hi, (n, ), generator=gen(n + hi)).to(dtype)
    s, p = pga.utils.index_sort(keys.to(dev), max_value=hi)
    rs, rp = O.index_sort(keys, stable=True)
    assert_close(s, rs)
    assert_close(p, rp)
    assert p.dtype == torch.int64
    s2, p2 = pga.utils.index_sort(keys.to(dev))  # unknown max_value: full-width radix
    assert_close(s2, rs)
    assert_close(p2, rp)


def test_index_sort_golden(dev, golden):
    import pytorch_geometric_amd as pga
    ix = golden['index']
    s, p = pga.utils.index_sort(ix['keys'].to(dev), max_value=37, stable=True)
    assert_close(s, ix['sorted'])
    assert_close(p, ix['perm'])
    assert_close(pga.index2ptr(s, 40), ix['ptr'])
    assert_close(pga.ptr2index(pga.index2ptr(s, 40)), ix['ptr2index'])
    p32 = pga.index2ptr(s.int(), 40)
    assert p32.dtype == torch.int32
    assert_close(p32, ix['keys32_ptr'])
    k = ix['known_index2ptr']
    assert pga.index2ptr(k['index'].to(dev), 3).tolist() == [0, 1, 3, 4]
    assert pga.index2ptr(k['index'].to(dev)).tolist() == [0, 1, 3, 4]  # size inferred


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_index2ptr_ptr2index_edge_cases(dev, dtype):
    import pytorch_geometric_amd as pga
    for idx, size in [([], 0), ([], 4), ([3, 3, 3], 6), ([0, 5], 6), ([2], 3)]:
        t = torch.tensor(idx, dtype=dtype)
        ref = O.index2ptr(t, size)
        got = pga.index2ptr(t.to(dev), size)
        assert_close(got, ref)
        assert got.dtype == dtype
        assert_close(pga.ptr2index(got, len(idx)), O.ptr2index(ref, len(idx)))


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_edge_index_handles(dev, dtype, golden):
    import pytorch_geometric_amd as pga
    e = golden['index']['known_edge_index']  # test/test_edge_index.py:196-233
    h = pga.EdgeIndex(e['edge_index'].to(dtype).to(dev), (3, 3)).fill_cache_()
    assert h.by_src().ptr.tolist() == [0, 1, 3, 4]
    assert h.by_dst().ptr.tolist() == [0, 1, 3, 4]
    assert h.by_dst().perm.tolist() == [1, 0, 3, 2]  # the stable one of the two accepted perms
    ei = random_graph(700, 900, 20_000, seed=5, dtype=dtype, skew=True)
    h = pga.EdgeIndex(ei.to(dev), (700, 900))
    for csr, key, other, n in [(h.by_dst(), ei[1], ei[0], 900), (h.by_src(), ei[0], ei[1], 700)]:
        ptr, idx, perm = O.csr_from_coo(key, other, n)
        assert_close(csr.ptr, ptr)
        assert_close(csr.idx, idx)
        assert_close(csr.perm.long(), perm)
        assert csr.ptr.dtype == dtype and csr.idx.dtype == dtype
    # sizes inferred from the data (maybe_num_nodes)
    h2 = pga.EdgeIndex(ei.to(dev))
    assert h2.sparse_size == (int(ei.max()) + 1, ) * 2
    m = h.src_slot_to_dst_slot()
    fwd, bwd = h.by_dst(), h.by_src()
    assert torch.equal(fwd.perm[m.long()], bwd.perm)


def test_edge_index_handle_is_a_tensor(dev):
    """``class EdgeIndex(Tensor)`` in the reference (edge_index.py:173): the handle can be indexed,
    passed to torch functions and to this package's own layers / utilities like the plain
    ``[2, E]`` tensor it shares storage with; operations return plain tensors."""
    import copy
    import pickle

    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.nn import GATConv, GCNConv, SAGEConv
    from pytorch_geometric_amd.utils import sort_edge_index
    ei = random_graph(300, 300, 4000, seed=3).to(dev)
    h = pga.EdgeIndex(ei, (300, 300))
    assert isinstance(h, torch.Tensor) and h.is_cuda and h.shape == ei.shape
    assert h.data_ptr() == ei.data_ptr() and h.as_tensor() is ei
    assert type(h[0]) is torch.Tensor and torch.equal(h[1], ei[1])
    assert type(h + 1) is torch.Tensor and torch.equal(torch.stack([h[0], h[1]]), ei)
    assert torch.equal(sort_edge_index(h, num_nodes=300), sort_edge_index(ei, num_nodes=300))
    h32 = h.to(torch.int32)
    assert isinstance(h32, pga.EdgeIndex) and h32.sparse_size == (300, 300)
    assert h32.dtype == torch.int32 and h.to(dev) is h
    assert type(h.to(torch.float32)) is torch.Tensor
    for twin in (copy.deepcopy(h), pickle.loads(pickle.dumps(h))):
        assert isinstance(twin, pga.EdgeIndex) and twin.sparse_size == h.sparse_size
        assert torch.equal(twin, ei)
    x = torch.randn(300, 16, device=dev)
    torch.manual_seed(0)
    for conv in (SAGEConv(16, 8), GCNConv(16, 8), GATConv(16, 4, heads=2)):
        conv = conv.to(dev)
        assert_close(conv(x, h), conv(x, ei), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError, match='shape'):
        pga.EdgeIndex(torch.zeros(3, 4, dtype=torch.long, device=dev))
    with pytest.raises(ValueError, match='data type'):
        pga.EdgeIndex(torch.zeros(2, 4, device=dev))


def test_presorted_edge_index(dev):
    import pytorch_geometric_amd as pga
    ei = random_graph(50, 60, 1000, seed=7)
    order = ei[1].sort(stable=True).indices
    ei_c = ei[:, order].contiguous()
    h = pga.EdgeIndex(ei_c.to(dev), (50, 60), sort_order='col')
    ptr, idx, _ = O.csr_from_coo(ei_c[1], ei_c[0], 60)
    assert_close(h.by_dst().ptr, ptr)
    assert_close(h.by_dst().idx, idx)


def test_hub_plan(dev):
    from pytorch_geometric_amd import _native
    deg = torch.tensor([0, 5, 2049, 1024, 1025, 0, 7000, 3])
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)]).to(dev)
    rows, cptr, n_hub, n_chunks = _native.hub_plan(ptr, threshold=1024, chunk=1024)
    assert n_hub == 3 and rows.tolist() == [2, 4, 6]
    assert cptr.tolist() == [0, 3, 5, 12] and n_chunks == 12
    assert _native.hub_plan(ptr, threshold=10_000, chunk=1024)[2] == 0
    big = torch.full((3000, ), 40, dtype=torch.int32)  # more hubs than one scan tile
    ptr = torch.cat([torch.zeros(1, dtype=torch.int32), big.cumsum(0).int()]).to(dev)
    rows, cptr, n_hub, n_chunks = _native.hub_plan(ptr, threshold=16, chunk=16)
    assert n_hub == 3000 and n_chunks == 9000
    assert cptr.tolist() == list(range(0, 9001, 3))


def test_index_minmax(dev):
    from pytorch_geometric_amd import _native
    t = torch.randint(3, 100_000, (1_000_003, ), generator=gen(3))
    assert _native.index_minmax(t.to(dev)) == (int(t.min()), int(t.max()))
    assert _native.index_minmax(t.int().to(dev)) == (int(t.min()), int(t.max()))
    assert _native.index_minmax(torch.empty(0, dtype=torch.long, device=dev))[1] == -1


def test_reference_accessor_names_and_matmul(dev):
    """get_csr / get_csc / sort_by / matmul with the reference's EdgeIndex semantics
    (test/test_edge_index.py:360, 733-775, 995-1052: every route equals gather+scatter)."""
    import pytorch_geometric_amd as pga
    ei = random_graph(60, 80, 1200, seed=3)
    h = pga.EdgeIndex(ei.to(dev), (60, 80))
    (rowptr, col), perm = h.get_csr()
    ptr, idx, p = O.csr_from_coo(ei[0], ei[1], 60)
    assert_close(rowptr, ptr); assert_close(col, idx); assert_close(perm, p)
    (colptr, row), perm = h.get_csc()
    ptr, idx, p = O.csr_from_coo(ei[1], ei[0], 80)
    assert_close(colptr, ptr); assert_close(row, idx); assert_close(perm, p)
    for order, key in (('row', 0), ('col', 1)):
        s, perm = h.sort_by(order)
        ref_perm = ei[key].sort(stable=True).indices
        assert_close(perm, ref_perm)
        assert_close(s.edge_index, ei[:, ref_perm])
        assert s.sort_by(order)[1] is None
    g = gen(8)
    x_cols = torch.randn(80, 9, generator=g)   # A is [60, 80]: A @ x needs 80 rows
    x_rows = torch.randn(60, 9, generator=g)
    val = torch.rand(1200, generator=g)
    for red in ('sum', 'mean', 'max'):
        ref = O.scatter(x_cols[ei[1]], ei[0], 0, 60, red)
        assert_close(h.matmul(x_cols.to(dev), reduce=red), ref, atol=2e-5, what=f'A@x {red}')
        ref_t = O.scatter(x_rows[ei[0]], ei[1], 0, 80, red)
        assert_close(h.matmul(x_rows.to(dev), reduce=red, transpose=True), ref_t, atol=2e-5)
    ref = O.scatter(x_cols[ei[1]] * val.view(-1, 1), ei[0], 0, 60, 'sum')
    xc = x_cols.to(dev).requires_grad_(True)
    out = h.matmul(xc, input_value=val.to(dev))
    assert_close(out, ref, atol=2e-5)
    out.sum().backward()
    ref_g = O.scatter(val.view(-1, 1).expand(-1, 9), ei[1], 0, 80, 'sum')
    assert_close(xc.grad, ref_g, atol=2e-5)


# ---- sort_edge_index / coalesce / to_undirected --------------------------------
def test_preprocessing_matches_reference_goldens(dev, golden_preproc):
    from pytorch_geometric_amd import utils as U
    from tests import _preproc_cases as P
    to = lambda t: t.to(dev)  # noqa: E731
    P.check_sort_edge_index(U, golden_preproc['sort_edge_index'], to)
    P.check_coalesce(U, golden_preproc['coalesce'], to)
    P.check_undirected(U, golden_preproc['undirected'], to)
    C = golden_preproc['coalesce']
    n = C['num_nodes']
    # call forms: no attr argument -> edge_index alone; nothing to merge -> attributes reordered
    assert torch.equal(U.coalesce(to(C['dup']), num_nodes=n).cpu(), C['no_attr'])
    S = golden_preproc['sort_edge_index']
    ei, a = U.coalesce(to(S['simple']), to(S['attr_f']), n)
    assert torch.equal(ei.cpu(), C['simple']['edge_index'])
    assert torch.equal(a.cpu(), C['simple']['attr'])
    assert torch.equal(U.to_undirected(to(golden_preproc['undirected']['edge_index'])).cpu(),
                       golden_preproc['undirected']['no_attr'])
    # gradient of merged attributes
    leaf = to(C['attr']).requires_grad_(True)
    _, a = U.coalesce(to(C['dup']), leaf, n, reduce='mean')
    (g, ) = torch.autograd.grad(a, leaf, to(C['mean_grad']['grad_out']))
    assert_close(g, C['mean_grad']['grad_attr'], atol=1e-6, what='coalesce grad')


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
def test_preprocessing_matches_oracle_at_size(dev, dtype):
    from pytorch_geometric_amd import utils as U
    n, E = 5000, 300_000
    ei = random_graph(n, n, E, seed=21, skew=True)          # heavy duplicates on the hubs
    w = torch.rand(E, generator=gen(22))
    for by_row in (True, False):
        got = U.sort_edge_index(ei.to(dtype).to(dev), num_nodes=n, sort_by_row=by_row)
        want, _ = O.sort_edge_index(ei, None, n, by_row)
        assert got.dtype == dtype and torch.equal(got.cpu().long(), want)
        gi, gw = U.coalesce(ei.to(dtype).to(dev), w.to(dev), n, sort_by_row=by_row)
        wi, ww = O.coalesce(ei, w, n, 'sum', False, by_row)
        assert torch.equal(gi.cpu().long(), wi)
        assert_close(gw, ww, atol=1e-5, what='merged weights')
    # the stable device sort keeps the input order inside a run: attributes match a stable sort
    gi, ga = U.sort_edge_index(ei.to(dev), torch.arange(E, device=dev), n)
    wi, wa = O.sort_edge_index(ei, torch.arange(E), n)
    assert torch.equal(ga.cpu(), wa)
    und = U.to_undirected(ei.to(dev), num_nodes=n)
    assert U.is_undirected(und, num_nodes=n) and not U.is_undirected(ei.to(dev), num_nodes=n)
    assert torch.equal(und.cpu(), O.to_undirected(ei, None, n)[0])


def test_preprocessing_edge_cases(dev):
    from pytorch_geometric_amd import utils as U
    empty = torch.empty(2, 0, dtype=torch.long, device=dev)
    assert U.sort_edge_index(empty, num_nodes=4).shape == (2, 0)
    ei, a = U.coalesce(empty, torch.empty(0, 3, device=dev), 4)
    assert ei.shape == (2, 0) and a.shape == (0, 3)
    pair = (torch.tensor([2, 0, 2], device=dev), torch.tensor([1, 3, 1], device=dev))
    out = U.coalesce(pair, num_nodes=4)                       # tuple in -> tuple out
    assert isinstance(out, tuple) and out[0].tolist() == [0, 2] and out[1].tolist() == [3, 1]
    one = torch.tensor([[3], [3]], device=dev)
    assert U.coalesce(one).tolist() == [[3], [3]]
    with pytest.raises(ValueError):
        U.coalesce(one, num_nodes=2**32)                      # key overflow, like the reference
    with pytest.raises(ValueError):
        U.sort_edge_index(torch.zeros(3, 4, dtype=torch.long, device=dev))


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('n', [0, 1, 63, 4096, 4097, 32768, 32769, 1_000_003])
def test_cumsum_bit_exact(dev, dtype, n):
    """``pygamd_cumsum`` = ``torch.cumsum(x, 0)`` on integer counts (one launch up to 32 k
    elements, three above), also in place and into the ``offsets[1:]`` slice the samplers use."""
    from pytorch_geometric_amd import _native
    x = torch.randint(0, 50, (n, ), generator=gen(n + 1)).to(dtype)
    want = torch.cumsum(x, 0)
    assert_close(_native.cumsum(x.to(dev)), want)
    offsets = torch.zeros(n + 1, dtype=dtype, device=dev)
    _native.cumsum(x.to(dev), out=offsets[1:])
    assert_close(offsets[1:], want)
    assert int(offsets[0]) == 0
    y = x.to(dev)
    assert _native.cumsum(y, out=y) is y
    assert_close(y, want)
    with pytest.raises(ValueError, match='one-dimensional'):
        _native.cumsum(torch.zeros(2, 2, dtype=dtype, device=dev))

"""BASELINE.json's full sizes (ogbn-products shape: N = 2,449,029, E = 61,859,140), checked through
size-independent properties — the oracle cannot run at this size in seconds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def products(dev):
    import pytorch_geometric_amd as pga
    from pytorch_geometric_amd.datasets import products_like
    x, y, ei, c = products_like(seed=1, scale=1.0)
    ei = ei.to(dev)
    h = pga.EdgeIndex(ei, (x.size(0), x.size(0)))
    return x.to(dev), ei, h


def test_csr_build_invariants_full_size(products):
    x, ei, h = products
    N, E = x.size(0), ei.size(1)
    for csr, key, other in ((h.by_dst(), ei[1], ei[0]), (h.by_src(), ei[0], ei[1])):
        ptr, idx, perm = csr.ptr, csr.idx, csr.perm
        assert ptr.numel() == N + 1 and int(ptr[0]) == 0 and int(ptr[-1]) == E
        assert bool((ptr[1:] >= ptr[:-1]).all())                      # monotone
        skey = key[perm]
        assert bool((skey[1:] >= skey[:-1]).all())                    # sortedness
        same = skey[1:] == skey[:-1]
        assert bool((perm[1:][same] > perm[:-1][same]).all())         # stability
        assert torch.equal(idx, other[perm])                          # payload follows the perm
        assert torch.equal(torch.bincount(key, minlength=N), ptr[1:] - ptr[:-1])  # ptr = counts
        assert int(perm.sum()) == E * (E - 1) // 2                    # a permutation of 0..E-1
    # idempotence: sorting the sorted keys is the identity permutation
    import pytorch_geometric_amd as pga
    skey = ei[1][h.by_dst().perm]
    s2, p2 = pga.utils.index_sort(skey, max_value=N)
    assert torch.equal(s2, skey) and torch.equal(p2, torch.arange(E, device=p2.device))
    assert h.by_dst().hub[2] > 0  # the power-law graph has hub rows (> 1024 in-edges)


@pytest.mark.parametrize('F', [100, 256])
def test_spmm_properties_full_size(products, F):
    import pytorch_geometric_amd as pga
    x, ei, h = products
    N, E = x.size(0), ei.size(1)
    g = torch.Generator(device=x.device).manual_seed(F)
    a = torch.randn(N, F, device=x.device, generator=g)
    b = torch.randn(N, F, device=x.device, generator=g)
    out_a = pga.utils.spmm(h, a, 'sum')
    # conservation: sum_i out[i] = sum_e a[src_e] = sum_j outdeg(j) * a[j]   (checked in fp64)
    outdeg = (h.by_src().ptr[1:] - h.by_src().ptr[:-1]).double()
    want = (a.double() * outdeg.view(-1, 1)).sum(0)
    got = out_a.double().sum(0)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-2), (got - want).abs().max()
    # linearity: A(2a - 3b) = 2 A a - 3 A b
    out_b = pga.utils.spmm(h, b, 'sum')
    out_c = pga.utils.spmm(h, 2 * a - 3 * b, 'sum')
    # (rounding scales with the sum of |terms| of a row, up to ~16 k terms on hub rows)
    err = (out_c - (2 * out_a - 3 * out_b)).abs()
    scale = pga.utils.spmm(h, 2 * a.abs() + 3 * b.abs(), 'sum').clamp(min=1.0)
    assert float((err / scale).max()) < 2e-6
    # mean = sum / clamp(deg, 1); rows without in-edges are exactly 0
    indeg = h.by_dst().degree()
    mean = pga.utils.spmm(h, a, 'mean')
    ref = out_a / indeg.clamp(min=1).to(torch.float32).view(-1, 1)
    assert float((mean - ref).abs().max()) < 1e-5
    assert bool((mean[indeg == 0] == 0).all())
    # adjoint: <A a, b> = <a, A^T b>  (forward vs the transposed kernel used in the backward)
    av = a.clone().requires_grad_(True)
    (pga.utils.spmm(h, av, 'sum') * b).sum().backward()
    lhs = (out_a.double() * b.double()).sum()
    rhs = (a.double() * av.grad.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * abs(float(lhs)) + 1.0
    # determinism: no atomics on the CSR path -> bitwise repeatable
    assert torch.equal(pga.utils.spmm(h, a, 'sum'), out_a)


def test_max_aggregation_full_size(products):
    import pytorch_geometric_amd as pga
    x, ei, h = products
    a = torch.randn(x.size(0), 64, device=x.device)
    mx = pga.utils.spmm(h, a, 'max')
    mn = pga.utils.spmm(h, -a, 'min')
    assert torch.equal(mx, -mn)                                        # max(a) = -min(-a)
    mean = pga.utils.spmm(h, a, 'mean')
    nonempty = h.by_dst().degree() > 0
    assert bool((mx[nonempty] >= mean[nonempty] - 1e-5).all())          # max >= mean
    # every reported maximum is the value of some in-neighbour (sampled rows)
    fwd = h.by_dst()
    rows = torch.randint(0, x.size(0), (200, )).tolist()
    for r in rows:
        s, e = int(fwd.ptr[r]), int(fwd.ptr[r + 1])
        if e > s:
            assert torch.equal(mx[r], a[fwd.idx[s:e]].max(0).values)

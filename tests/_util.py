import torch

RTOL = 1e-5
ATOL = 1e-5  # north_star: aggregation within 1e-5 fp32 of PyG's CPU scatter path


def assert_close(got, ref, rtol=RTOL, atol=ATOL, what=''):
    got = got.detach().cpu()
    ref = ref.detach().cpu()
    assert got.shape == ref.shape, f'{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}'
    if got.dtype.is_floating_point:
        err = (got - ref).abs()
        tol = atol + rtol * ref.abs()
        bad = err > tol
        assert not bad.any(), (f'{what}: {int(bad.sum())} / {got.numel()} elements off, '
                               f'max abs err {float(err.max()):.3e}')
    else:
        assert torch.equal(got, ref), f'{what}: integer mismatch'


def gen(seed):
    return torch.Generator().manual_seed(seed)


def random_graph(n_src, n_dst, n_edges, seed, dtype=torch.int64, skew=False):
    g = gen(seed)
    src = torch.randint(0, n_src, (n_edges, ), generator=g)
    if skew:  # a few hub destinations
        dst = (torch.rand(n_edges, generator=g).pow(6) * n_dst).long().clamp(max=n_dst - 1)
    else:
        dst = torch.randint(0, n_dst, (n_edges, ), generator=g)
    return torch.stack([src, dst]).to(dtype)


def assert_sum_close(got, ref32, exact64, rtol=RTOL, atol=ATOL, what='', abs_sum=None):
    """For long fp32 sums neither the CPU reference nor the kernel is exact, and two summation
    orders legitimately differ by more than 1e-5 on hub rows (thousands of terms).  Judge both
    against the float64 result: the kernel must be within 1e-5 (relative) of the exact value OR at
    least as close to it as twice the reference's own worst error."""
    got = got.detach().cpu().double()
    ref32 = ref32.detach().cpu().double()
    exact64 = exact64.detach().cpu().double()
    assert got.shape == exact64.shape, f'{what}: shape {tuple(got.shape)} vs {tuple(exact64.shape)}'
    err = (got - exact64).abs()
    ref_err = float((ref32 - exact64).abs().max()) if ref32.numel() else 0.0
    tol = torch.clamp(atol + rtol * exact64.abs(), min=2 * ref_err)
    if abs_sum is not None:
        # condition-aware bound for sums with cancellation: a few fp32 ulps of sum_i |x_i|
        tol = torch.maximum(tol, 1e-6 * abs_sum.detach().cpu().double())
    bad = err > tol
    assert not bad.any(), (f'{what}: {int(bad.sum())} / {got.numel()} elements off, max err vs '
                           f'fp64 {float(err.max()):.3e} (reference fp32 err {ref_err:.3e})')

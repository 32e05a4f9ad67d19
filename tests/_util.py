import torch

RTOL = 1e-5
ATOL = 1e-5  # north_star: aggregation within 1e-5 fp32 of PyG's CPU scatter path


def assert_close(got, ref, rtol=RTOL, atol=ATOL, what=''):
    got = got.detach().cpu()
    ref = ref.detach().cpu()
    assert got.shape == ref.shape, f'{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}'
    if got.dtype.is_floating_point:
        # isclose, not `err > tol`: a NaN/Inf in `got` where `ref` is finite must FAIL (NaN
        # compares False with everything); non-finite reference entries must be reproduced
        # exactly (same-signed inf, NaN for NaN)
        ok = torch.isclose(got, ref.to(got.dtype), rtol=rtol, atol=atol, equal_nan=True)
        if not bool(ok.all()):
            bad = ~ok
            err = (got - ref).abs()
            finite = err[torch.isfinite(err)]
            worst = float(finite.max()) if finite.numel() else float('nan')
            nonfinite = int((~torch.isfinite(got) & torch.isfinite(ref)).sum())
            raise AssertionError(f'{what}: {int(bad.sum())} / {got.numel()} elements off '
                                 f'({nonfinite} non-finite where the reference is finite), '
                                 f'max finite abs err {worst:.3e}')
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
    assert bool(torch.isfinite(exact64).all()), f'{what}: the fp64 evaluation is not finite'
    bad = ~(err <= tol)  # NaN-proof: a non-finite `got` is never <= tol
    assert not bad.any(), (f'{what}: {int(bad.sum())} / {got.numel()} elements off, max err vs '
                           f'fp64 {float(err.nan_to_num(nan=float("inf")).max()):.3e} '
                           f'(reference fp32 err {ref_err:.3e})')


def assert_close_scaled(got, ref, tol=2e-5, what=''):
    """Deep-model gradients: error relative to the largest magnitude of the reference tensor (per
    element rtol is meaningless where large terms cancel)."""
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape, f'{what}: shape'
    assert bool(torch.isfinite(got).all()) or not bool(torch.isfinite(ref).all()), \
        f'{what}: non-finite values where the reference is finite'
    err = float((got - ref).abs().max()) if got.numel() else 0.0
    scale = max(float(ref.abs().max()), 1.0) if ref.numel() else 1.0
    assert err <= tol * scale, f'{what}: max abs err {err:.3e} vs scale {scale:.3e}'


def assert_close_outliers(got, ref, tol=2e-5, max_outlier_frac=2e-3, what=''):
    """Gradients through ReLU stacks at large N: an activation within one ulp of 0 can land on
    different sides on the two devices, which flips its mask and perturbs the gradients of a few
    neighbouring rows by O(|grad|).  Require the bulk to match at `tol` (relative to the tensor's
    scale) and bound the number of such rows."""
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape, f'{what}: shape'
    assert bool(torch.isfinite(got).all()) or not bool(torch.isfinite(ref).all()), \
        f'{what}: non-finite values where the reference is finite'
    scale = max(float(ref.abs().max()), 1.0)
    bad = ~((got - ref).abs() <= tol * scale)  # NaN-proof
    frac = float(bad.float().mean())
    assert frac <= max_outlier_frac, f'{what}: {frac:.2e} of the elements differ by > {tol:g}'
    assert float((got - ref).abs().max()) <= 0.05 * scale, f'{what}: gross mismatch'


def decompress_rows(z, F):
    """CPU reading of the compressed-row layout (include/pyg_amd.h, pygamd_rows_compress): int32
    [n, ld] -> (float32 [n, F], kept mask [n, F]); row = 8 mask words + kept values in order."""
    import numpy as np
    zz = z.detach().cpu().numpy().view(np.uint32)
    n = zz.shape[0]
    cols = np.arange(F)
    mask = ((zz[:, cols >> 5] >> (cols & 31).astype(np.uint32)) & 1).astype(bool)
    out = np.zeros((n, F), dtype=np.uint32)
    for i in range(n):
        k = int(mask[i].sum())
        out[i, mask[i]] = zz[i, 8:8 + k]
    return torch.from_numpy(out.view(np.float32)), torch.from_numpy(mask)


def assert_close_rows(got, ref, tol=2e-5, max_bad_rows=8, what=''):
    """Gradients behind a kinked activation evaluated ONCE per edge (GAT's leaky_relu on
    alpha_src[j] + alpha_dst[i]): a pre-activation within fp32 rounding of 0 takes the other slope
    on the other device — expected about once per 1e7 evaluations — and moves the gradient rows of
    that edge's two end points by O(|grad|).  Every element must match at `tol` (relative to the
    tensor's scale) except in at most `max_bad_rows` rows, and those must stay within 5 % of the
    scale."""
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape, f'{what}: shape'
    assert bool(torch.isfinite(got).all()), f'{what}: non-finite values'
    scale = max(float(ref.abs().max()), 1.0)
    err = (got - ref).abs()
    bad_rows = int((~(err <= tol * scale)).reshape(err.size(0), -1).any(1).sum())
    assert bad_rows <= max_bad_rows, (f'{what}: {bad_rows} rows differ by > {tol:g} x scale '
                                      f'(max abs err {float(err.max()):.3e}, scale {scale:.3e})')
    assert float(err.max()) <= 0.05 * scale, f'{what}: gross mismatch {float(err.max()):.3e}'

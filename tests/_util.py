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

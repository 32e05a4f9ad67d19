"""Synthetic graphs with the SHAPES of the reference's benchmark datasets (no network: the real
files cannot be downloaded).  Generated on the CPU from a seeded ``torch.Generator`` so the CPU
oracle and the GPU see identical bits."""
from typing import Tuple

import torch
from torch import Tensor

# name -> (num_nodes, num_edges, num_features, num_classes)
SHAPES = {
    'cora': (2_708, 10_556, 1_433, 7),                 # datasets/planetoid.py:65-69
    'ogbn-arxiv': (169_343, 1_166_243, 128, 40),
    'ogbn-products': (2_449_029, 61_859_140, 100, 47),
    'fb15k-237': (14_541, 544_230, 0, 474),            # datasets/rel_link_pred_dataset.py:39-40
}


def powerlaw_undirected(num_nodes: int, num_edges: int, seed: int, alpha: float = 0.54,
                        dtype: torch.dtype = torch.int64, device=None) -> Tensor:
    r"""``[2, num_edges]`` edge list of an undirected graph stored in both directions (like
    ogbn-products): ``num_edges / 2`` pairs ``(u, v)`` with ``u`` drawn from a Zipf-like popularity
    ``p(rank) ~ rank^-alpha`` over randomly permuted node ids and ``v`` uniform, then mirrored.
    ``alpha = 0.54`` gives a maximum degree of about 17 k at the ogbn-products shape.
    ``device``: generate there (a device generator has its own stream of numbers — the same
    distribution, not the same bits; for graphs too large to build on the host, e.g. the
    ogbn-papers100M shape with 1.6 G edges)."""
    device = torch.device('cpu') if device is None else torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed)
    half = num_edges // 2
    w = torch.arange(1, num_nodes + 1, dtype=torch.float64, device=device).pow_(-alpha)
    cdf = w.cumsum(0)
    cdf /= cdf[-1].clone()
    del w
    edge_index = torch.empty(2, num_edges, dtype=dtype, device=device)
    r = torch.rand(half, generator=g, dtype=torch.float64, device=device)
    u = torch.searchsorted(cdf, r).clamp_(max=num_nodes - 1)
    del r, cdf
    u = torch.randperm(num_nodes, generator=g, device=device)[u]
    v = torch.randint(0, num_nodes, (half, ), generator=g, device=device)
    edge_index[0, :half], edge_index[0, half:2 * half] = u, v
    edge_index[1, :half], edge_index[1, half:2 * half] = v, u
    if num_edges % 2 == 1:  # one extra directed edge to hit an odd count exactly
        edge_index[:, -1] = torch.randint(0, num_nodes, (2, ), generator=g, device=device)
    return edge_index


def uniform_directed(num_nodes: int, num_edges: int, seed: int,
                     dtype: torch.dtype = torch.int64) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, num_nodes, (2, num_edges), generator=g).to(dtype)


def products_like(seed: int = 1, scale: float = 1.0, skewed: bool = True,
                  dtype: torch.dtype = torch.int64) -> Tuple[Tensor, Tensor, Tensor, int]:
    """(x [N,100], y [N], edge_index [2,E], num_classes) at ``scale`` x the ogbn-products shape."""
    n, e, f, c = SHAPES['ogbn-products']
    n, e = max(int(n * scale), 2), max(int(e * scale) // 2 * 2, 2)
    ei = (powerlaw_undirected if skewed else uniform_directed)(n, e, seed, dtype=dtype)
    g = torch.Generator().manual_seed(seed + 1_000_003)
    x = torch.randn(n, f, generator=g)
    y = torch.randint(0, c, (n, ), generator=g)
    return x, y, ei, c

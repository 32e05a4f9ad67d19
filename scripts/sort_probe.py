"""index_sort (csrc/graph.hip: own stable LSD radix sort, 8 bits per pass) against torch.sort(stable=True)
at the edge counts of the bench graph: time and bit-equality of keys and permutation.
Usage: python scripts/sort_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device=dev).manual_seed(0)
print('n, max key, dtype: own ms / torch.sort(stable) ms, equal keys, equal permutation')
for n, hi, dtype in [(61_859_140, 2_449_029, torch.int64), (61_859_140, 2_449_029, torch.int32),
                     (1_000_000, 170_000, torch.int64), (10_000_000, 2 ** 40, torch.int64),
                     (5_000, 100, torch.int64)]:
    key = (torch.rand(n, device=dev, generator=g) ** 3 * hi).to(dtype).clamp_(max=hi - 1)
    own = timeit(lambda: _native.index_sort(key, max_value=hi - 1))
    ref = timeit(lambda: torch.sort(key, stable=True))
    ks, perm = _native.index_sort(key, max_value=hi - 1)
    rk, rp = torch.sort(key, stable=True)
    print(f'{n:10d}, {hi:14d}, {str(dtype):11s}: {own:8.3f} / {ref:8.3f}, {torch.equal(ks, rk)}, '
          f'{torch.equal(perm, rp)}', flush=True)

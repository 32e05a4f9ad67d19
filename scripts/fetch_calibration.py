"""Calibration of rocprofv3's FETCH_SIZE for rows shorter than 1 KiB:
a plain row gather with a KNOWN byte count — every source row read exactly once through a random
permutation — at row pitches of 1024 B (F = 256), 400 B (F = 100), 192 B (F = 48) and 512 B
(F = 100 stored at a 128-float pitch).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`;
prints the algorithmic bytes of each launch (rows + indices read) and the 128-byte lines the rows
touch, in launch order."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

dev = torch.device('cuda:0')
N = 2_000_000
g = torch.Generator().manual_seed(0)
perm = torch.randperm(N, generator=g).to(dev)
for F, pitch in ((256, 256), (100, 100), (48, 48), (100, 128)):
    buf = torch.randn(N, pitch, device=dev)
    x = buf[:, :F]
    for _ in range(3):
        out = _native.gather_rows(x, perm)
    torch.cuda.synchronize()
    row_b, pitch_b = 4 * F, 4 * pitch
    # 128-byte lines touched by row r: [r * pitch_b, r * pitch_b + row_b)
    r = torch.arange(N, dtype=torch.int64)
    lines = ((r * pitch_b + row_b - 1) // 128 - (r * pitch_b) // 128 + 1).sum().item()
    print(f'F={F} pitch={pitch}: rows read {N * row_b / 1e9:.3f} GB + index {N * 8 / 1e9:.3f} GB; '
          f'128-byte lines touched by the rows: {lines * 128 / 1e9:.3f} GB; 3 launches', flush=True)

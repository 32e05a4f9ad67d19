"""The production split weight gradient on random, constant and zero inputs (same launches, same
instruction stream): how much of its time is the chip clocking down under full-toggle operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

rows, K1, K2, N = 2_449_029, 256, 256, 256
dev = torch.device('cuda:0')
_native.set_gemm_mode('split')
out = torch.empty(N, K1 + K2, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, make in (('randn', lambda *s: torch.randn(*s, device=dev)),
                   ('ones', lambda *s: torch.ones(*s, device=dev)),
                   ('zeros', lambda *s: torch.zeros(*s, device=dev))):
    x, x2, go = make(rows, K1), make(rows, K2), make(rows, N)
    parts = []
    for variant in (0, 8, 12, 10):
        _native.lab_set_wgrad_variant(variant)
        parts.append(f'variant {variant}: {timeit(lambda: _native.linear_wgrad(go, x, out=out, bias_grad=True, x2=x2)):.3f}')
    _native.lab_set_wgrad_variant(0)
    print(name, ', '.join(parts), flush=True)
    del x, x2, go

"""gemm_nt (forward / dgrad) at the shapes of configs 3 and 2: ms, fp32-equivalent TFLOP/s, and the
bytes each launch has to move.  Usage: python scripts/nt_shapes_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=20):
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


for mode in (('split', ) if os.environ.get('ONLY_SPLIT') else ('split', 'fp32')):
    _native.set_gemm_mode(mode)
    print('mode', mode)
    for M, K, N in ((169_343, 128, 256), (169_343, 256, 256), (169_343, 256, 320),
                    (169_343, 320, 256), (2_449_029, 96, 256), (2_449_029, 256, 48),
                    (2_449_029, 256, 256), (14_541, 500, 500), (16_384, 256, 256), (1_024, 256, 256),
                    (100_000, 100, 256)):
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.05
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: _native.linear_forward(x, w, None, out=out))
        t_lib = timeit(lambda: torch.mm(x, w.t(), out=out))
        gb = (M * K + M * N + N * K) * 4 / 1e9
        print(f'  [{M}, {K}] x [{N}, {K}]^T: {t:7.3f} ms  {2.0 * M * K * N / t / 1e9:6.1f} TF/s  '
              f'{gb:5.2f} GB -> {gb / t:5.2f} TB/s   (library {t_lib:7.3f} ms)', flush=True)
        del x, w, out

"""Weight-gradient (TN) kernels of csrc/gemm.hip at the three layer shapes of the bench step
(ogbn-products: M = 2,449,029 rows; [agg | x]^T @ grad with the bias gradient from the same pass)
and at sampled-block sizes: the fp32 instruction, round 3's split schedule (operands split in
registers; lab switch) and the production split schedule (operands split once into LDS), against
the library (torch.mm).  Usage: python scripts/wgrad_probe.py [--rows M]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=2_449_029)
ap.add_argument('--skinny-only', action='store_true', help='only the narrow-g A/B at the end')
args = ap.parse_args()
dev = torch.device('cuda:0')
M = args.rows
# (rows, K1, K2, N): layer 1 (100 | 100 -> 256), layer 2 (256 | 256 -> 256), layer 3 (-> 47)
SHAPES = [(M, 100, 100, 256), (M, 256, 256, 256), (M, 256, 256, 47), (16384, 256, 256, 256),
          (180224, 100, 100, 256)]


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


g = torch.Generator(device=dev).manual_seed(0)
print('rows, K1 | K2 -> N: fp32 / split in registers (r3) / split once (production) / library ms'
      ' ; production TFLOP/s ; max |production - r3|')
for rows, K1, K2, N in ([] if args.skinny_only else SHAPES):
    x = torch.randn(rows, K1, device=dev, generator=g)
    x2 = torch.randn(rows, K2, device=dev, generator=g)
    go = torch.randn(rows, N, device=dev, generator=g)
    out = torch.empty(N, K1 + K2, device=dev)
    cat = torch.cat([x, x2], 1)

    def own():
        return _native.linear_wgrad(go, x, out=out, bias_grad=True, x2=x2)

    ts, res = [], []
    for mode, variant in (('fp32', 0), ('split', 1), ('split', 0)):
        prev = _native.set_gemm_mode(mode)
        _native.lab_set_wgrad_variant(variant)
        ts.append(timeit(own))
        res.append(own()[0].clone())
        _native.lab_set_wgrad_variant(0)
        _native.set_gemm_mode(prev)
    lib = timeit(lambda: torch.mm(go.t(), cat, out=out))
    flops = 2.0 * rows * (K1 + K2) * N
    print(f'{rows:8d}, {K1} | {K2} -> {N}: {ts[0]:7.3f} / {ts[1]:7.3f} / {ts[2]:7.3f} / {lib:7.3f}'
          f' ; {flops / ts[2] / 1e9:6.1f} ; {float((res[2] - res[1]).abs().max()):.1e}',
          flush=True)
    if rows == M and N % 4 == 0:  # where the time goes: phases switched off (lab probes)
        prev = _native.set_gemm_mode('split')
        parts = []
        for bits, what in ((2, 'no products'), (4, 'no conversion'), (8, 'no loads'),
                           (6, 'loads only'), (12, 'products only'), (10, 'conversion only'),
                           (14, 'barriers only'), (32, 'packed subtractions'),
                           (40, 'packed subtractions, no loads')):
            _native.lab_set_wgrad_variant(bits)
            parts.append(f'{what} {timeit(own):.3f}')
        _native.lab_set_wgrad_variant(0)
        _native.set_gemm_mode(prev)
        print('          phases off: ' + ', '.join(parts), flush=True)
    del x, x2, go, cat

# ---- round 6: gemm_tn_skinny_kernel (narrow g, N <= 96 — the bench step's classifier layer, g =
# [A^T g' | g'] 96 columns wide against ONE 256-column x) against the tiled kernel (lab switch 16)
print('rows, K1 | K2 -> N: tiled split kernel / streamed kernel ms ; GB/s of the streamed one'
      ' ; max |difference|')
for rows, K, K2, N in [(M, 256, 0, 96), (M, 512, 0, 48), (M, 256, 0, 64), (180224, 256, 0, 96),
                       (65536, 256, 0, 96)]:
    x = torch.randn(rows, K, device=dev, generator=g)
    x2 = torch.randn(rows, K2, device=dev, generator=g) if K2 else None
    go = torch.randn(rows, N, device=dev, generator=g)
    out = torch.empty(N, K + K2, device=dev)

    def own():
        return _native.linear_wgrad(go, x, out=out, bias_grad=True, x2=x2)

    prev = _native.set_gemm_mode('split')
    _native.lab_set_wgrad_variant(16)
    t_old = timeit(own)
    r_old = own()[0].clone()
    _native.lab_set_wgrad_variant(0)
    t_new = timeit(own)
    r_new = own()[0].clone()
    _native.set_gemm_mode(prev)
    gbs = rows * (K + K2 + N) * 4 / t_new / 1e6
    print(f'{rows:8d}, {K} | {K2} -> {N}: {t_old:7.3f} / {t_new:7.3f} ; {gbs:7.0f} ; '
          f'{float((r_old - r_new).abs().max()):.3e}', flush=True)
    if rows == M and N == 96 and K == 256:
        # phases of the streamed kernel switched off (lab probes 64 + bits)
        prev = _native.set_gemm_mode('split')
        parts = []
        for bits, what in ((2, 'no products'), (4, 'no conversion'), (8, 'no loads'),
                           (12, 'products only'), (10, 'conversion only')):
            _native.lab_set_wgrad_variant(64 + bits)
            parts.append(f'{what} {timeit(own):.3f}')
        _native.lab_set_wgrad_variant(0)
        _native.set_gemm_mode(prev)
        print('          phases off: ' + ', '.join(parts), flush=True)
        # the same launch on all-zero and on constant operands: what the data's toggling costs
        for name, fill in (('zeros', 0.0), ('ones', 1.0)):
            x.fill_(fill)
            go.fill_(fill)
            if x2 is not None:
                x2.fill_(fill)
            _native.set_gemm_mode('split')
            print(f'          {name}: streamed {timeit(own):.3f}', end='')
            _native.lab_set_wgrad_variant(16)
            print(f', tiled {timeit(own):.3f}', flush=True)
            _native.lab_set_wgrad_variant(0)
            _native.set_gemm_mode(prev)
    del x, x2, go

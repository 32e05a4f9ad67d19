"""Times the fp32-MFMA dense transform (csrc/gemm.hip) against torch.mm (rocBLAS / hipBLASLt, with
and without the shipped TunableOp table) on the eight GEMM shapes of the ogbn-products step.
Usage: python scripts/gemm_probe.py [--rows N] [--no-tuned]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=2_449_029)
ap.add_argument('--no-tuned', action='store_true')
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--mode', default='fp32', choices=['fp32', 'split'])
ap.add_argument('--only', default='', help='comma list of fwd,dgrad,wgrad')
ap.add_argument('--pad', type=int, default=0, help='extra floats in the leading dimension of the activations')
args = ap.parse_args()
if not args.no_tuned:
    from pytorch_geometric_amd.tuning import enable_tuned_gemms
    print('tuned table:', enable_tuned_gemms())
dev = torch.device('cuda:0')
_native.set_gemm_mode(args.mode)
print('gemm mode:', _native.get_gemm_mode())
only = set(args.only.split(',')) if args.only else {'fwd', 'dgrad', 'wgrad'}
M = args.rows


def timeit(fn, reps=args.reps):
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


def report(name, flops, t_own, t_lib, err):
    print(f'{name:34s} own {t_own:7.3f} ms {flops / t_own / 1e9:6.1f} TF | lib {t_lib:7.3f} ms '
          f'{flops / t_lib / 1e9:6.1f} TF | max rel diff {err:.2e}', flush=True)


g = torch.Generator(device=dev).manual_seed(0)
for K, N in ((200, 256), (512, 256), (256, 96)) if 'fwd' in only else ():
    x = torch.randn(M, K + args.pad, device=dev, generator=g)[:, :K]
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    out = torch.empty(M, N + args.pad, device=dev)[:, :N]
    ref = torch.empty(M, N + args.pad, device=dev)[:, :N]
    t_own = timeit(lambda: _native.linear_forward(x, w, b, relu=True, out=out))
    t_lib = timeit(lambda: torch.relu_(torch.addmm(b, x, w.t(), out=ref)))
    err = float((out - ref).abs().max() / ref.abs().max())
    report(f'fwd  [M,{K}]x[{N},{K}]^T+b,relu', 2.0 * M * K * N, t_own, t_lib, err)
    del x, out, ref
for N, K in ((256, 512), (96, 256)) if 'dgrad' in only else ():
    go = torch.randn(M, N, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    wt = w.t().contiguous()
    scale = torch.rand(M, device=dev, generator=g)
    out = torch.empty(M, K, device=dev)
    t_own = timeit(lambda: _native.linear_dgrad(go, wt, scale, K // 2, out=out))
    ref = torch.empty(M, K, device=dev)
    t_lib = timeit(lambda: torch.mm(go, w, out=ref))
    ref[:, :K // 2] *= scale.view(-1, 1)
    err = float((out - ref).abs().max() / ref.abs().max())
    report(f'dgrad [M,{N}]x[{N},{K}] (+row scale)', 2.0 * M * K * N, t_own, t_lib, err)
    del go, out, ref
for N, K in ((256, 200), (256, 512), (96, 256)) if 'wgrad' in only else ():
    go = torch.randn(M, N, device=dev, generator=g)
    x = torch.randn(M, K, device=dev, generator=g)
    out = torch.empty(N, K, device=dev)
    t_own = timeit(lambda: _native.linear_wgrad(go, x, out=out))
    ref = torch.empty(N, K, device=dev)
    t_lib = timeit(lambda: torch.mm(go.t(), x, out=ref))
    err = float((out - ref).abs().max() / ref.abs().max())
    report(f'wgrad [M,{N}]^T x [M,{K}]', 2.0 * M * K * N, t_own, t_lib, err)
    del go, x

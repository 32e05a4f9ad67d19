"""Times the one-kernel SAGE layer (csrc/sage_fused.hip) at the ogbn-products shape: both gather
variants, the weight-prefetch depths, and — through the kernel's probe bits — its phases in
isolation (gather loop skipped / MFMA loop skipped), next to the stand-alone SpMM and GEMM.
Usage: python scripts/fused_probe.py [--scale s]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_geometric_amd as pga  # noqa: E402
from pytorch_geometric_amd import _native  # noqa: E402
from pytorch_geometric_amd.datasets import products_like  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--widths', default='256,100')
ap.add_argument('--only-spec', action='store_true', help='only the producer/consumer variants')
ap.add_argument('--prio', action='store_true', help='A/B of the issue-priority scheme')
ap.add_argument('--pitch', action='store_true', help='F = 100 rows at a 128-float pitch')
ap.add_argument('--split', action='store_true',
                help='only: production fp32 kernel (6) vs production split kernel (5) + its phases')
ap.add_argument('--sq-only', action='store_true',
                help='a few launches of v1 and v3 with phases on/off (for a rocprofv3 --pmc pass)')
args = ap.parse_args()
dev = torch.device('cuda:0')
x0, _, ei, _ = products_like(seed=1, scale=args.scale)
N = x0.size(0)
h = pga.EdgeIndex(ei.to(dev), (N, N))
fwd = h.by_dst()
fwd.hub


def timeit(fn, reps=8):
    for _ in range(2):
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
for F in [int(v) for v in args.widths.split(',')]:
    Fo = 256
    buf = torch.empty(N, 2 * F, device=dev)
    buf[:, F:] = torch.randn(N, F, device=dev, generator=g)
    xsrc = buf[:, F:].contiguous() if F == 100 else buf[:, F:]
    w = torch.randn(Fo, 2 * F, device=dev, generator=g) * 0.05
    b = torch.randn(Fo, device=dev, generator=g)
    out = torch.empty(N, Fo, device=dev)
    ref = torch.empty(N, Fo, device=dev)

    def one(variant, probe=0, save_agg=True):
        _native.SAGE_FUSED_PROBE = probe
        _native.sage_layer_forward(fwd.ptr, fwd.idx, xsrc, buf[:, F:], w, b, 'mean', True,
                                   buf[:, :F], out, hub=fwd.hub, save_agg=save_agg,
                                   variant=variant)
        _native.SAGE_FUSED_PROBE = 0

    t_spmm = timeit(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, xsrc, 'mean', n_rows=N,
                                             hub=fwd.hub, out=buf[:, :F]))
    t_gemm = timeit(lambda: _native.linear_forward(buf, w, b, relu=True, out=ref))
    print(f'F={F} Fo={Fo}: SpMM {t_spmm:.3f} ms, GEMM {t_gemm:.3f} ms', flush=True)
    if args.split:
        one(6)
        err6 = float((out - ref).abs().max() / ref.abs().max())
        one(5)
        err5 = float((out - ref).abs().max() / ref.abs().max())
        print(f'  fp32 kernel {timeit(lambda: one(6)):.3f} ms (max rel diff vs GEMM {err6:.1e}); '
              f'split kernel {timeit(lambda: one(5)):.3f} ms ({err5:.1e}); split: gather skipped '
              f'{timeit(lambda: one(5, 1)):.3f} ms, matrix loop skipped '
              f'{timeit(lambda: one(5, 2)):.3f} ms, both {timeit(lambda: one(5, 3)):.3f} ms; '
              f'agg not stored {timeit(lambda: one(5, 0, False)):.3f} ms; three workgroups per '
              f'CU (two-pass kernel, ring of 2) {timeit(lambda: one(5, 16)):.3f} ms; fp32 v1 gather skipped '
              f'{timeit(lambda: one(1, 1)):.3f} ms, MFMA skipped {timeit(lambda: one(1, 2)):.3f} ms',
              flush=True)
        continue
    if args.pitch and F == 100:
        # layer 1's rows are 400 bytes: 3.125 lines each, straddling — against the same rows stored
        # at a 512-byte pitch (whole lines, 28 % more bytes)
        wide = torch.zeros(N, 128, device=dev)
        wide[:, :F] = xsrc
        xw = wide[:, :F]

        def one_wide(variant=1, probe=0):
            _native.SAGE_FUSED_PROBE = probe
            _native.sage_layer_forward(fwd.ptr, fwd.idx, xw, xw, w, b, 'mean', True, buf[:, :F], out,
                                       hub=fwd.hub, save_agg=True, variant=variant)
            _native.SAGE_FUSED_PROBE = 0

        print(f'  F=100, v1: 400-byte pitch {timeit(lambda: one(1)):.3f} ms (MFMA skipped '
              f'{timeit(lambda: one(1, 2)):.3f}); 512-byte pitch {timeit(one_wide):.3f} ms (MFMA '
              f'skipped {timeit(lambda: one_wide(1, 2)):.3f}); stand-alone SpMM 400 / 512: '
              f'{t_spmm:.3f} / ' + f'''{timeit(lambda: _native.spmm_csr(fwd.ptr, fwd.idx, xw, 'mean', n_rows=N, hub=fwd.hub, out=buf[:, :F])):.3f} ms''',
              flush=True)
        continue
    if args.prio:
        for variant in (1, 3, 4):
            print(f'  v{variant}: default {timeit(lambda: one(variant)):.3f} ms, gather at s_setprio 2 '
                  f'{timeit(lambda: one(variant, 2048)):.3f} ms; MFMA skipped '
                  f'{timeit(lambda: one(variant, 2)):.3f} / {timeit(lambda: one(variant, 2048 | 2)):.3f}'
                  f' ms', flush=True)
        continue
    if args.sq_only:
        for variant in (1, 3):
            for probe in (0, 1, 2):   # full / gather skipped / MFMA skipped: told apart by order
                for _ in range(3):
                    one(variant, probe)
        torch.cuda.synchronize()
        continue
    for variant in (3, 4):
        out.fill_(float('nan'))
        one(variant)
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f'  v{variant} (producer/consumer, {4 if variant == 3 else 8} transform waves) '
              f'{timeit(lambda: one(variant)):.3f} ms   (max rel diff {err:.1e}); '
              f'agg not stored {timeit(lambda: one(variant, 0, False)):.3f} ms; '
              f'gather skipped {timeit(lambda: one(variant, 1)):.3f} ms, MFMA skipped '
              f'{timeit(lambda: one(variant, 2)):.3f} ms, both {timeit(lambda: one(variant, 3)):.3f} ms; '
              f'2 buffers {timeit(lambda: one(variant, 2 << 8)):.3f} ms, 3 buffers '
              f'{timeit(lambda: one(variant, 3 << 8)):.3f} ms; static roles '
              f'{timeit(lambda: one(variant, 16)):.3f} ms, gather skipped '
              f'{timeit(lambda: one(variant, 17)):.3f} ms', flush=True)
        print(f'     hot-address probes: weights {timeit(lambda: one(variant, 32)):.3f} ms, root '
              f'{timeit(lambda: one(variant, 64)):.3f} ms, both {timeit(lambda: one(variant, 96)):.3f} ms, '
              f'no epilogue {timeit(lambda: one(variant, 128)):.3f} ms, all three '
              f'{timeit(lambda: one(variant, 224)):.3f} ms; gather skipped + all three '
              f'{timeit(lambda: one(variant, 225)):.3f} ms', flush=True)
    if args.only_spec:
        continue
    one(1)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'  v1 (row-at-a-time)      {timeit(lambda: one(1)):.3f} ms   (max rel diff {err:.1e})',
          flush=True)
    for pf in (1, 2, 3):
        one(2, pf << 2)
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f'  v2 (streamed) PF={pf}      {timeit(lambda: one(2, pf << 2)):.3f} ms   '
              f'(max rel diff {err:.1e})', flush=True)
    print(f'  v2 PF=2, agg not stored {timeit(lambda: one(2, 8, False)):.3f} ms', flush=True)
    for variant in (1, 2):
        print(f'  v{variant}: gather loop skipped {timeit(lambda: one(variant, 8 | 1)):.3f} ms, '
              f'MFMA loop skipped {timeit(lambda: one(variant, 8 | 2)):.3f} ms, '
              f'both skipped {timeit(lambda: one(variant, 8 | 3)):.3f} ms', flush=True)
    for variant in (1, 2):
        print(f'  v{variant}: no weight loads after the first chunk(s) '
              f'{timeit(lambda: one(variant, 8 | 16)):.3f} ms, no LDS fragment reads '
              f'{timeit(lambda: one(variant, 8 | 32)):.3f} ms, neither '
              f'{timeit(lambda: one(variant, 8 | 48)):.3f} ms; gather skipped + no weight loads '
              f'{timeit(lambda: one(variant, 8 | 16 | 1)):.3f} ms', flush=True)
    for variant in (1, 2):
        print(f'  v{variant}, ONE workgroup per CU: full {timeit(lambda: one(variant, 8 | 64)):.3f} ms, '
              f'MFMA loop skipped {timeit(lambda: one(variant, 8 | 64 | 2)):.3f} ms, gather skipped '
              f'{timeit(lambda: one(variant, 8 | 64 | 1)):.3f} ms', flush=True)
    for variant in (1, 2):
        print(f'  v{variant}, two accumulation chains: full {timeit(lambda: one(variant, 8 | 128)):.3f} ms, '
              f'gather skipped {timeit(lambda: one(variant, 8 | 128 | 1)):.3f} ms', flush=True)

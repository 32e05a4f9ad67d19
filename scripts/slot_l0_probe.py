"""Phases of the one-kernel SAGE layer at the shape of config 4's FIRST layer (slot batches:
169,984 destination rows owning 5 / 10 / 15 slots each, neighbours read from the graph's feature
matrix by int64 node id, F = 128 -> 256): the production split kernel whole, with its gather loop
skipped (probe bit 0) and with its matrix loop skipped (probe bit 1), through the laboratory
library.  Usage: python scripts/slot_l0_probe.py [--nodes 20000000] [--fill 0.75]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=20_000_000)
ap.add_argument('--fill', type=float, default=0.75, help='share of rows that are not holes')
ap.add_argument('--F', type=int, default=128)
ap.add_argument('--Fo', type=int, default=256)
args = ap.parse_args()
dev = torch.device('cuda:0')
B, fan = 1024, [15, 10, 5]
cap = [B]
for k in fan:
    cap.append(cap[-1] * k)
R_dst = sum(cap[:3])
g = torch.Generator(device=dev).manual_seed(0)
F, Fo = args.F, args.Fo
x = torch.randn(args.nodes, F, device=dev, generator=g)
begin, end, off = [], [], 0
for b, k in enumerate(fan):
    s = off + torch.arange(cap[b], device=dev, dtype=torch.int64) * k
    live = torch.rand(cap[b], device=dev, generator=g) < args.fill
    begin.append(s)
    end.append(torch.where(live, s + k, s))
    off += cap[b] * k
row_begin, row_end = torch.cat(begin), torch.cat(end)
S = off
src = torch.randint(0, args.nodes, (S, ), device=dev, generator=g)
cat = torch.randn(R_dst, 2 * F, device=dev, generator=g)
w = torch.randn(Fo, 2 * F, device=dev, generator=g) * 0.05
bias = torch.randn(Fo, device=dev, generator=g)
out = torch.empty(R_dst, Fo, device=dev)
bits = _native.relu_bits_like(R_dst, Fo, dev)


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
    return e0.elapsed_time(e1) / reps * 1e3


def one(probe, variant=5):
    _native.SAGE_FUSED_PROBE = probe
    _native.sage_layer_forward(row_begin, src, x, cat[:, F:], w, bias, 'mean', True, cat[:, :F],
                               out, save_agg=True, relu_bits=bits, rowend=row_end,
                               variant=variant)
    _native.SAGE_FUSED_PROBE = 0


print(f'rows {R_dst}, slots {S}, F {F} -> {Fo}, fill {args.fill}')
timeit(lambda: one(0), reps=300)   # clocks up (the first timings of a fresh process read ~15 % high)
for name, probe in (('whole', 0), ('no gather', 1), ('no matrix loop', 2), ('neither', 3)):
    print(f'  split kernel, {name:16s} {timeit(lambda: one(probe)):8.1f} us')
print(f'  fp32-instruction kernel       {timeit(lambda: one(0, 6)):8.1f} us')

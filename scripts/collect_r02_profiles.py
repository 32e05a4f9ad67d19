"""Copies the round-2 rocprofv3 results from gpurun_out/ (scratch) into profiles/ (tracked):
kernel-stats tables of the bench / the mini-batch mode / the min-max probe, and the PMC traffic
of the dominant kernels as JSON.  Usage: python scripts/collect_r02_profiles.py <tag>"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import summarize_profile  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
N, E, F = 2449029, 61859140, 256

for name, steps, title in (
        ('bench', 7, 'rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 '
                     '--no-cpu-baseline (round 2; 7 steps, MI355X)'),
        ('bench_split', 7, 'PYGAMD_GEMM_MODE=split rocprofv3 --kernel-trace --stats -- python bench.py '
                           '--steps 5 --warmup 2 --no-cpu-baseline (round 2; the opt-in 3 x bf16 split '
                           'arithmetic of the stand-alone GEMM kernels; 7 steps)'),
        ('minibatch', 50, 'rocprofv3 --kernel-trace --stats -- python bench.py --mode minibatch '
                          '--steps 40 --warmup 10 (round 2; full papers100M shape, 50 batches)'),
        ('minmax', 1, 'rocprofv3 --kernel-trace --stats -- python scripts/reduce_probe.py (round 2; '
                      'sum/mean/max/min aggregation fwd+bwd at the products shape, F = 256; '
                      '4 calls per direction)')):
    src = os.path.join(G, f'prof_{tag}_{name}', 'trace_kernel_stats.csv')
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, f'r02_{name}_kernel_stats.csv'))
        summarize_profile.main(src, os.path.join(P, f'r02_{name}_kernel_stats.md'), title, steps)
        print('wrote', f'profiles/r02_{name}_kernel_stats.md')


def counters(kind):
    path = os.path.join(G, f'pmc_{tag}_{kind}', 'summary.txt')
    out = {}
    if os.path.exists(path):
        for line in open(path):
            parts = line.split()
            out[' '.join(parts[4:])] = float(parts[2])
    return out


fetch, write = counters('fetch'), counters('write')
if fetch and write:
    def pick(d, sub):
        ks = [k for k in d if sub in k]
        return d[ks[0]] if ks else None

    res = {
        'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, '
                  '--kernel-trace only), python bench.py --steps 2 --warmup 1 --no-cpu-baseline on '
                  'MI355X, round 2 (scripts/gpu_r02_profile.sh)',
        'workload': {'scale': 1.0, 'index_dtype': 'int64', 'graph': 'power-law', 'N': N, 'E': E,
                     'F': F},
        'units': 'FETCH_SIZE / WRITE_SIZE are KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE '
                 'on gfx950 reports exactly half of the bytes of a wide (16 B/lane) coalesced '
                 'read, so it is doubled',
        'per_launch': {},
    }
    for key, sub in (('spmm_sum_rows_F256_transposed_accumulate', 'spmm_sum_rows<long, 4, 64, 1, 0'),
                     ('sage_fused_fwd_F256', 'sage_fused_fwd_kernel<long, 64>'),
                     ('sage_fused_fwd_F100', 'sage_fused_fwd_kernel<long, 32>'),
                     ('spmm_sum_rows_F48', 'spmm_sum_rows<long, 4, 16, 1, 0'),
                     ('gemm_tn_wgrad', 'gemm_tn_kernel<true, false>'),
                     ('gemm_nt_128x128', 'gemm_nt_kernel<2, 2, 2, 2, true, false>')):
        f, w = pick(fetch, sub), pick(write, sub)
        if f is not None and w is not None:
            res['per_launch'][key] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
                                      'hbm_bytes': (2 * f + w) * 1024}
    # + the rows accumulated onto + the one-bit-per-element ReLU mask of the fused ReLU backward
    alg = E * (4 * F + 8) + (N + 1) * 8 + N * 4 * F + N * 4 * F + N * 4 * (F // 32)
    dom = res['per_launch'].get('spmm_sum_rows_F256_transposed_accumulate')
    if dom:
        res['kernel'] = 'pygamd::spmm_sum_rows<long,4,64,1,0,false> (transposed, accumulate, ReLU-backward epilogue)'
        res['algorithmic_bytes_per_launch'] = alg
        res['traffic_bytes_per_launch'] = dom['hbm_bytes']
        res['traffic_over_algorithmic'] = round(dom['hbm_bytes'] / alg, 4)
    with open(os.path.join(P, 'r02_pmc_spmm_f256.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print('wrote profiles/r02_pmc_spmm_f256.json', res.get('traffic_over_algorithmic'))

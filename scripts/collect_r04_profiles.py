"""Copies the round-4 rocprofv3 results from gpurun_out/ (scratch; scripts/gpu_r04_profile.sh) into
profiles/ (tracked): kernel-stats tables of the bench (split arithmetic = the default, and the exact
fp32 instruction), of the captured / eager slot-batch mini-batch mode and of configs 3 / 5, and the
PMC traffic of the dominant kernels as JSON.  Usage: python scripts/collect_r04_profiles.py"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import summarize_profile  # noqa: E402

G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
N, E = 2449029, 61859140

for name, steps, title in (
        ('bench', 7, 'rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 '
                     '--no-cpu-baseline --no-side-figures (round 4; default arithmetic = 3 x bf16 '
                     'split; 7 steps, MI355X)'),
        ('bench_fp32', 7, 'rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 '
                          '--no-cpu-baseline --no-side-figures --arith fp32 (round 4; the exact '
                          'fp32 matrix instruction everywhere; 7 steps)'),
        ('minibatch', 120, 'rocprofv3 --kernel-trace --stats -- python bench.py --mode minibatch '
                           '--capture --steps 100 --warmup 20 (round 4; slot batches, one hipGraph '
                           'per batch, full papers100M shape; 120 replays + 3 eager warm-ups + '
                           'the graph build)'),
        ('minibatch_eager', 63, 'rocprofv3 --kernel-trace --stats -- PYGAMD_CAPTURE=0 python '
                                'bench.py --mode minibatch --capture --steps 50 --warmup 10 (round 4; '
                                'the same static-shape step launched eagerly: per-kernel times; 63 '
                                'batches + the graph build)'),
        ('config3', 13, 'rocprofv3 --kernel-trace --stats -- python scripts/time_gat.py (BASELINE '
                        'config 3, GAT 3-layer heads=8, ogbn-arxiv shape; round 4; 13 steps)'),
        ('config5', 13, 'rocprofv3 --kernel-trace --stats -- python scripts/time_rgcn.py (BASELINE '
                        'config 5, RGCNConv(500, 500, 474, num_blocks=5) x 2, FB15k-237 shape; '
                        'round 4; 13 steps)')):
    src = os.path.join(G, f'prof_r04_{name}', 'trace_kernel_stats.csv')
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, f'r04_{name}_kernel_stats.csv'))
        summarize_profile.main(src, os.path.join(P, f'r04_{name}_kernel_stats.md'), title, steps)
        print('wrote', f'profiles/r04_{name}_kernel_stats.md')
        log = os.path.join(G, f'prof_r04_{name}', 'stdout.log')
        if os.path.exists(log):
            lines = [l for l in open(log) if l.startswith('{') or l.startswith('config')]
            if lines:
                with open(os.path.join(P, f'r04_{name}_kernel_stats.md'), 'a') as f:
                    f.write('\nProgram output under the profiler: `' + lines[-1].strip()[:600]
                            + '`\n')


def counters(kind):
    path = os.path.join(G, f'pmc_r04_{kind}', 'summary.txt')
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

    src = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace '
           'only), python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures on '
           'MI355X, round 4 (scripts/gpu_r04_profile.sh), default arithmetic (split)')
    units = ('FETCH_SIZE / WRITE_SIZE are KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on '
             'gfx950 reports exactly half of the bytes of a wide (16 B/lane) coalesced read, so it '
             'is doubled (calibration: profiles/r03_fetch_size_calibration.txt).')
    by_symbol, detail = {}, {}
    for sym, sub in (('sage_fused_split_kernel<long,64>', 'sage_fused_split_kernel<long, 64'),
                     ('sage_fused_split_kernel<long,32>', 'sage_fused_split_kernel<long, 32'),
                     ('sage_fused_fwd_kernel<long,64>', 'sage_fused_fwd_kernel<long, 64'),
                     ('sage_fused_fwd_kernel<long,32>', 'sage_fused_fwd_kernel<long, 32'),
                     ('spmm_sum_rows<long,F=48>', 'spmm_sum_rows<long, 4, 16, 1, 0'),
                     ('spmm_sum_rows_sparse<long,F=48>', 'spmm_sum_rows_sparse<long, 4, 16, 1'),
                     ('gemm_tn_split_kernel', 'gemm_tn_split_kernel'),
                     ('gemm_tn_kernel', 'gemm_tn_kernel<'), ('gemm_nt_kernel', 'gemm_nt_kernel'),
                     ('rows_pack_kernel', 'rows_pack_kernel')):
        fv, wv = pick(fetch, sub), pick(write, sub)
        if fv is not None and wv is not None:
            by_symbol[sym] = (2 * fv + wv) * 1024
            detail[sym] = {'FETCH_SIZE_KiB': fv, 'WRITE_SIZE_KiB': wv}
    F = 256
    # CHANGELOG.md §3 / bench.py fused_algorithmic_bytes: the forward launch (aggregated rows stored) and
    # the input-gradient launch (row-scaled second output instead); the counters average over both
    fwd = E * (4 * F + 8) + (N + 1) * 8 + 3 * N * 4 * F + N * 4 * (F // 32)
    bwd = E * (4 * F + 8) + (N + 1) * 8 + 3 * N * 4 * F + N * 4 * (F // 32)
    alg = (fwd + bwd) / 2
    extra = {}
    key = 'sage_fused_split_kernel<long,64>'
    if key in by_symbol:
        extra = {'dominant_kernel': key, 'algorithmic_bytes_per_launch_mean': alg,
                 'traffic_over_algorithmic': round(by_symbol[key] / alg, 4),
                 'note': 'average over the forward launches (gather + root rows + stored '
                         'aggregated rows + output + ReLU bits) and the input-gradient launches '
                         '(gather + root rows + output + its row-scaled copy + mask bits); the '
                         'weight term planes (768 KiB per launch) stay in L2'}
    with open(os.path.join(P, 'r04_pmc_bench.json'), 'w') as f:
        json.dump({'source': src, 'units': units,
                   'workload': {'scale': 1.0, 'index_dtype': 'int64', 'graph': 'power-law', 'N': N,
                                'E': E},
                   'traffic_bytes_per_launch': by_symbol, 'counters': detail, **extra}, f, indent=1)
    print('wrote profiles/r04_pmc_bench.json', extra.get('traffic_over_algorithmic'))

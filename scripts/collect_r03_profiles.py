"""Copies the round-3 rocprofv3 results from gpurun_out/ (scratch; scripts/gpu_r03_profile.sh) into
profiles/ (tracked): kernel-stats tables of the bench / the min-max probe / the mini-batch mode,
the PMC traffic of the dominant kernels as JSON, the SQ counters of the fused-layer variants per
phase probe.  Usage: python scripts/collect_r03_profiles.py"""
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
                     '--no-cpu-baseline (round 3; 7 steps, MI355X)'),
        ('minmax', 1, 'rocprofv3 --kernel-trace --stats -- python scripts/reduce_probe.py (round 3; '
                      'sum/mean/max/min aggregation fwd+bwd at the products shape, F = 256; 4 calls '
                      'per direction; the min/max backward = minmax_pack_kernel + '
                      'minmax_bwd_src_kernel + spmm_minmax_bwd_dst for the marked outputs)'),
        ('minibatch', 50, 'rocprofv3 --kernel-trace --stats -- python bench.py --mode minibatch '
                          '--steps 40 --warmup 10 (round 3; eager path with 2 prefetched batches, '
                          'full papers100M shape, 50 batches)')):
    src = os.path.join(G, f'prof_r03_{name}', 'trace_kernel_stats.csv')
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, f'r03_{name}_kernel_stats.csv'))
        summarize_profile.main(src, os.path.join(P, f'r03_{name}_kernel_stats.md'), title, steps)
        print('wrote', f'profiles/r03_{name}_kernel_stats.md')


def counters(kind):
    path = os.path.join(G, f'pmc_r03_{kind}', 'summary.txt')
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
                  'MI355X, round 3 (scripts/gpu_r03_profile.sh)',
        'workload': {'scale': 1.0, 'index_dtype': 'int64', 'graph': 'power-law', 'N': N, 'E': E},
        'units': 'FETCH_SIZE / WRITE_SIZE are KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE '
                 'on gfx950 reports exactly half of the bytes of a wide (16 B/lane) coalesced '
                 'read, so it is doubled.  Calibrated on a known byte count at 1024 / 400 / 192-byte '
                 'row pitches (profiles/r03_fetch_size_calibration.txt): the doubled figure is the '
                 'line-granular (128-byte) traffic within +2...4 % at every pitch.',
        'per_launch': {},
    }
    for key, sub in (('sage_fused_fwd_F256 (layer-2 forward AND layer-2 input gradient)',
                      'sage_fused_fwd_kernel<long, 64'),
                     ('sage_fused_fwd_F100 (layer-1 forward)', 'sage_fused_fwd_kernel<long, 32'),
                     ('spmm_sum_rows_F48 (layer-3 forward)', 'spmm_sum_rows<long, 4, 16, 1, 0'),
                     ('spmm_sum_rows_sparse_F48 (layer-3 backward: zero rows of the loss '
                      'gradient skipped)', 'spmm_sum_rows_sparse<long, 4, 16, 1'),
                     ('rows_pack (layout pass of the loss gradient + live-row bitmap)',
                      'rows_pack_kernel'),
                     ('gemm_tn_wgrad', 'gemm_tn_kernel<true, false>'),
                     ('gemm_nt_128x128', 'gemm_nt_kernel<2, 2, 2, 2, true, false>')):
        f, w = pick(fetch, sub), pick(write, sub)
        if f is not None and w is not None:
            res['per_launch'][key] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
                                      'hbm_bytes': (2 * f + w) * 1024}
    F = 256
    # CHANGELOG.md §3: gathered rows + indices + row pointers + root rows + saved aggregated rows + output
    alg = E * (4 * F + 8) + (N + 1) * 8 + N * 4 * F + N * 4 * F + N * 4 * F
    dom = [v for k, v in res['per_launch'].items() if k.startswith('sage_fused_fwd_F256')]
    if dom:
        res['kernel'] = 'pygamd::sage_fused_fwd_kernel<long,64> (the dominant kernel of the step)'
        res['algorithmic_bytes_per_launch'] = alg
        res['traffic_bytes_per_launch'] = dom[0]['hbm_bytes']
        res['traffic_over_algorithmic'] = round(dom[0]['hbm_bytes'] / alg, 4)
    with open(os.path.join(P, 'r03_pmc_fused_f256.json'), 'w') as f:
        json.dump(res, f, indent=1)
    # the same numbers keyed by bench.py's kernel symbols (bench.py `pmc_traffic`)
    by_symbol = {}
    for sym, sub in (('sage_fused_fwd_kernel<long,64>', 'sage_fused_fwd_kernel<long, 64'),
                     ('sage_fused_fwd_kernel<long,32>', 'sage_fused_fwd_kernel<long, 32'),
                     ('spmm_sum_rows<long,F=48>', 'spmm_sum_rows<long, 4, 16, 1, 0'),
                     ('spmm_sum_rows_sparse<long,F=48>', 'spmm_sum_rows_sparse<long, 4, 16, 1')):
        fv, wv = pick(fetch, sub), pick(write, sub)
        if fv is not None and wv is not None:
            by_symbol[sym] = (2 * fv + wv) * 1024
    with open(os.path.join(P, 'r03_pmc_bench.json'), 'w') as f:
        json.dump({'source': res['source'], 'units': res['units'],
                   'workload': res['workload'], 'traffic_bytes_per_launch': by_symbol}, f,
                  indent=1)
    print('wrote profiles/r03_pmc_fused_f256.json', res.get('traffic_over_algorithmic'))

src = os.path.join(G, 'pmc_r03_sq_fused', 'dispatches.txt')
if os.path.exists(src):
    with open(os.path.join(P, 'r03_pmc_fused_sq_counters.txt'), 'w') as out:
        out.write(
            '# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY '
            'SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU '
            '--kernel-trace -- python scripts/fused_probe.py --widths 256 --sq-only (round 3, '
            'MI355X, products shape, F = 256 -> 256)\n'
            '# Every dispatch in order.  sage_fused_fwd_kernel (v1, production) and '
            'sage_fused_spec_kernel (v3, producer / consumer waves): 3 x full kernel, 3 x gather '
            'loop skipped, 3 x MFMA loop skipped.  spmm_sum_rows / gemm_nt: the stand-alone '
            'launches of the same work.\n'
            '# WAVE_CYCLES = WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: '
            'MFMA pipe / dependency / memory pipe full) + ACTIVE_INST_ANY, summed over all waves.\n')
        out.write(open(src).read())
    print('wrote profiles/r03_pmc_fused_sq_counters.txt')

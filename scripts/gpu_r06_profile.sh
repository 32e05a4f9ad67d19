#!/bin/bash
# Round-6 profiles in one GPU-box visit (results under gpurun_out/, summaries copied to profiles/ by
# scripts/collect_r06_profiles.py):
#   1. rocprofv3 --kernel-trace --stats of the default bench line (7 steps)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE — separate passes, kernel-trace only) of 3 steps
#   3. one PMC pass of SQ counters over the fused-layer variants (v1 = production, v3 = producer /
#      consumer waves) in scripts/fused_probe.py: where the wave cycles go
#   4. rocprofv3 --kernel-trace --stats of the min/max probe and of the eager mini-batch mode
set -u
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
TAG=${TAG:-r06}
R=$GRAFT_REPO_ROOT
prof() {  # prof <name> <cmd...>
  local OUT=$R/gpurun_out/prof_${TAG}_$1; shift
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- "$@" > $OUT/stdout.log 2>&1)
  grep -v "W2026\|E2026\|amdgpu.ids" $OUT/stdout.log | tail -2 | cut -c1-300
  find $OUT -name "*kernel_trace*.csv" -size +8M -delete
}
pmc() {  # pmc <name> <counters> <kernel filter> <cmd...>
  local OUT=$R/gpurun_out/pmc_${TAG}_$1; local CTR=$2; local FILT=$3; shift 3
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d $OUT -o pmc --output-format csv -- "$@" > $OUT/stdout.log 2>&1)
  python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*counter_collection*.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if re.search(r'$FILT', k):
            acc[(k[:90], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as out:
    for (k, c), v in sorted(acc.items()):
        line = f'{c:26s} avg {sum(v)/len(v):18.1f} x{len(v):3d}  {k}'
        print(line); out.write(line + '\n')
with open('$OUT/dispatches.txt', 'w') as out:  # every dispatch, in order (phase probes differ)
    for (k, c), v in sorted(acc.items()):
        out.write(f'{c:26s} {k}\n    ' + ' '.join(f'{x:.4g}' for x in v) + '\n')
PY
  find $OUT -name "*.csv" -size +8M -delete
}
# ONLY=traffic: just the two HBM-traffic passes (the rest of the visit costs 8 GPU-minutes)
want() { [[ -z "${ONLY:-}" || "$ONLY" == "$1" ]]; }
if want stats; then
echo "== kernel stats: bench"
prof bench python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-figures
fi
if want traffic; then
echo "== PMC FETCH_SIZE"
pmc fetch FETCH_SIZE 'spmm_sum_rows|sage_fused|gemm_|rows_pack' python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures
echo "== PMC WRITE_SIZE"
pmc write WRITE_SIZE 'spmm_sum_rows|sage_fused|gemm_|rows_pack' python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures
fi
if want configs; then
echo "== kernel stats: configs 3 / 5"
prof config3 python $R/scripts/time_gat.py
prof config5 python $R/scripts/time_rgcn.py
python $R/scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/${TAG}_configs_timings.txt
fi
if want minibatch; then
echo "== kernel stats: minibatch, captured slot batches (full papers100M shape)"
prof minibatch python $R/bench.py --mode minibatch --capture --steps 100 --warmup 20
echo "== kernel stats: minibatch, the same static-shape step eagerly (per-kernel times)"
PYGAMD_CAPTURE=0 prof minibatch_eager python $R/bench.py --mode minibatch --capture --steps 50 --warmup 10
fi
[[ -n "${ONLY:-}" ]] && exit 0
echo "== kernel stats: bench with the exact fp32 instruction (side figure)"
prof bench_fp32 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-figures --arith fp32
echo "== kernel stats: minibatch, captured slot batches (full papers100M shape)"
prof minibatch python $R/bench.py --mode minibatch --capture --steps 100 --warmup 20
echo "== kernel stats: minibatch, the same static-shape step eagerly (per-kernel times)"
PYGAMD_CAPTURE=0 prof minibatch_eager python $R/bench.py --mode minibatch --capture --steps 50 --warmup 10
echo "== kernel stats: configs 3 / 5"
prof config3 python $R/scripts/time_gat.py
prof config5 python $R/scripts/time_rgcn.py

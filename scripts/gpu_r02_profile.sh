#!/bin/bash
# Round-2 profiles in one GPU-box visit (results under gpurun_out/, summaries copied to profiles/
# by scripts/collect_r02_profiles.py afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default bench line (7 steps)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE — separate passes, kernel-trace only) of 3 steps
#   3. rocprofv3 --kernel-trace --stats of the mini-batch mode at the full papers100M shape
#   4. the min/max aggregation probe
set -u
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
prof() {  # prof <name> <cmd...>
  local OUT=$R/gpurun_out/prof_${TAG}_$1; shift
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- "$@" > $OUT/stdout.log 2>&1)
  grep -v "W2026\|E2026" $OUT/stdout.log | tail -2 | cut -c1-400
  find $OUT -name "*kernel_trace*.csv" -size +8M -delete
}
pmc() {  # pmc <name> <counter> <cmd...>
  local OUT=$R/gpurun_out/pmc_${TAG}_$1; local CTR=$2; shift 2
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 900 rocprofv3 --pmc $CTR --kernel-trace -d $OUT -o pmc --output-format csv -- "$@" > $OUT/stdout.log 2>&1)
  python - <<PY
import csv, glob, collections
files = glob.glob('$OUT/*counter_collection*.csv')
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'spmm_sum_rows' in k or 'sage_fused' in k or 'gemm_' in k:
            acc[(k[:90], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as out:
    for (k, c), v in sorted(acc.items()):
        line = f'{c:12s} avg {sum(v)/len(v):16.1f} x{len(v):3d}  {k}'
        print(line); out.write(line + '\n')
PY
  find $OUT -name "*.csv" -size +8M -delete
}
echo "== kernel stats: bench"
prof bench python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
echo "== PMC FETCH_SIZE"
pmc fetch FETCH_SIZE python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
echo "== PMC WRITE_SIZE"
pmc write WRITE_SIZE python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
echo "== kernel stats: minibatch (full papers100M shape)"
prof minibatch python $R/bench.py --mode minibatch --steps 40 --warmup 10 --no-cpu-baseline
if [ -f $R/scripts/reduce_probe.py ]; then
  echo "== min/max probe"
  prof minmax python $R/scripts/reduce_probe.py
fi

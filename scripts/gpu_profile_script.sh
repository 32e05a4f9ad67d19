#!/bin/bash
# rocprofv3 kernel stats of one script: bash scripts/gpu_profile_script.sh <tag> <script.py> [args]
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > $OUT/stdout.log 2>&1)
grep -v "rocprofv3\|W2026\|E2026" $OUT/stdout.log | tail -5
find $OUT -name "*kernel_trace*.csv" -size +20M -delete
python - <<PY
import csv, re
rows = list(csv.DictReader(open('$OUT/trace_kernel_stats.csv')))
for r in rows[:24]:
    n = r['Name'].replace('void ', '')
    if n.startswith('Cijk'): n = 'GEMM ' + (re.search(r'_(MT\d+x\d+x\d+)_', n) or [0,''])[1]
    print(f"{int(r['TotalDurationNs'])/1e6:9.2f} ms  x{r['Calls']:>5}  avg {float(r['AverageNs'])/1e3:9.1f} us  {n[:100]}")
PY

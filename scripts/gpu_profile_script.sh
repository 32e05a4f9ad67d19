#!/bin/bash
# usage: gpu_profile_script.sh <tag> <script.py> [args...]
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > $OUT/stdout.log 2>&1)
grep -v -E "amdgpu.ids|rocprofv3|^W2026|^E2026" $OUT/stdout.log | tail -4
python - <<PY
import csv, re
rows = list(csv.DictReader(open('$OUT/trace_kernel_stats.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.1f} ms')
for r in rows[:22]:
    n = r['Name'].replace('void ', '')
    if n.startswith('Cijk'): n = 'GEMM ' + (re.search(r'_(MT\d+x\d+x\d+)_', n) or [0,''])[1]
    print(f"{int(r['TotalDurationNs'])/1e6:9.2f} ms  x{r['Calls']:>5}  avg {float(r['AverageNs'])/1e3:9.1f} us  {n[:90]}")
PY

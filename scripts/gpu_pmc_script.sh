#!/bin/bash
# one rocprofv3 --pmc pass over a script: bash scripts/gpu_pmc_script.sh <tag> "<counters>" <kernel substr> <script.py>
set -u
export TMPDIR=/tmp
TAG=$1; CTRS=$2; KSUB=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && STEPS=3 timeout 900 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > $OUT/stdout.log 2>&1)
grep -v "W2026\|E2026" $OUT/stdout.log | tail -3
python - <<PY
import csv, glob, collections
files = glob.glob('$OUT/*counter_collection*.csv')
if not files:
    print('no counter file'); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    if '$KSUB' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f'   {c:32s} {sum(v)/len(v):18.1f} avg x{len(v)}')
PY
find $OUT -name "*.csv" -size +20M -delete

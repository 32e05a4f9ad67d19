#!/bin/bash
# One PMC pass over a command: gpu_pmc_one.sh <name> <counters> <kernel regex> <cmd...>
# (rocprofv3 --pmc with --kernel-trace only; averages per kernel name into gpurun_out/pmc_<name>/summary.txt)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1; CTR=$2; FILT=$3; shift 3
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d $OUT -o pmc --output-format csv -- "$@" > $OUT/stdout.log 2>&1)
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*counter_collection*.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if re.search(r'$FILT', k):
            acc[(k[:70], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as out:
    for (k, c), v in sorted(acc.items()):
        line = f'{c:22s} x{len(v):3d} ' + ' '.join(f'{x:.4g}' for x in v[:24]) + f'  {k}'
        print(line); out.write(line + '\n')
PY
find $OUT -name "*.csv" -size +8M -delete

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python scripts/fused_probe.py --widths 100 --pitch 2>&1 | grep -v amdgpu.ids
for CTR in FETCH_SIZE WRITE_SIZE; do
OUT=$R/gpurun_out/pmc_r03_calib_$CTR; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d $OUT -o pmc --output-format csv -- python $R/scripts/fetch_calibration.py > $OUT/stdout.log 2>&1)
grep "^F=" $OUT/stdout.log
python - <<PY
import csv, glob
vals=[]
for f in glob.glob('$OUT/*counter_collection*.csv'):
    for r in csv.DictReader(open(f)):
        if 'gather_rows_kernel' in r['Kernel_Name']:
            vals.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
vals.sort()
print('$CTR per gather_rows launch (KiB, launch order):', [round(v) for _, v in vals])
open('$OUT/summary.txt','w').write(' '.join(str(v) for _, v in vals))
PY
find $OUT -name "*.csv" -size +8M -delete
done

#!/bin/bash
# SQ counters of the dense transform kernels (one rocprofv3 --pmc pass, kernel-trace only) for both
# arithmetic modes.  Usage: gpurun -- 'bash scripts/gpu_pmc_gemm.sh <tag> [probe args]'
TAG=${1:-gemm}; shift
R=$(pwd); export TMPDIR=/tmp
CTRS=${CTRS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"}
for MODE in ${MODES:-split fp32}; do
  OUT=$R/gpurun_out/pmc_${TAG}_$MODE; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 600 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT -o pmc --output-format csv -- python $R/scripts/gemm_probe.py --mode $MODE --reps 2 "$@" > $OUT/stdout.log 2>&1)
  grep -v amdgpu.ids $OUT/stdout.log | grep "own" | cut -c1-120
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*counter_collection*.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'pygamd::gemm_' in k and 'reduce' not in k:
            acc[(k[:60], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as out:
    for (k, c), v in sorted(acc.items()):
        line = f'{c:26s} avg {sum(v)/len(v):16.1f} x{len(v):3d}  {k}'
        print(line); out.write(line + '\n')
PY
  find $OUT -name "*.csv" -size +8M -delete
done

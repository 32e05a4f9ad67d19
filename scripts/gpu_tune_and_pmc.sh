#!/bin/bash
# (1) TunableOp: let PyTorch pick the best rocBLAS/hipBLASLt solution per GEMM shape of the bench,
#     save the CSV, re-run the bench reading it.  (2) PMC passes (FETCH_SIZE / WRITE_SIZE) in
#     their own rocprofv3 runs, as the microarchitecture guide prescribes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
CSV=$GRAFT_REPO_ROOT/gpurun_out/tunableop_results.csv
rm -f $CSV
echo "== tuning run"
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$CSV \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 \
  timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
ls -la gpurun_out/tunableop_results*.csv 2>/dev/null; wc -l gpurun_out/tunableop_results*.csv 2>/dev/null
echo "== bench with tuned GEMMs"
F=$(ls gpurun_out/tunableop_results*.csv | head -1)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$F \
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_tuned.log | cut -c1-300
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$C
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d $OUT -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stdout.log 2>&1)
  ls $OUT | head
  python - <<PY
import csv, glob, collections
files = glob.glob('$OUT/*counter_collection*.csv')
if not files:
    print('no counter file'); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    if r.get('Counter_Name') == '$C':
        acc[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f'{sum(v)/len(v):16.1f} avg  x{len(v):3d}  {k}')
PY
done

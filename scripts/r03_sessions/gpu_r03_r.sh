#!/bin/bash
# compressed rows (zero-skipping gather source): layout / bit-exactness tests, the fused stack,
# smoke, the bench line with and without, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm.py tests/test_gpu_layers.py -m gpu -q -x \
  -k "compress or fused_sage_stack or training_split" > $O/r_tests.log 2>&1
rc=$?; echo "new tests rc=$rc"; tail -25 $O/r_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout 300 python __graft_entry__.py smoke > $O/r_smoke.log 2>&1
echo "smoke rc=$?"; grep "\[smoke\]" $O/r_smoke.log | cut -c1-400
for CR in 1 0; do
PYGAMD_COMPRESS_ROWS=$CR timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r_bench_cr$CR.json 2> $O/r_bench_cr$CR.err
echo "bench compress_rows=$CR rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r_bench_cr$CR.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('ms/step', round(d['ms_per_step'],2), r.get('kernel'), r.get('avg_launch_ms'), 'frac', r.get('frac'))
    print('others', r.get('others'))
except Exception as e:
    print('ERR', e)
PY
done
OUT=$O/prof_r; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stdout.log 2>&1)
echo "prof rc=$?"
python scripts/summarize_profile.py $(find $OUT -name "*kernel_stats.csv" | head -1) $O/r_bench_kernel_stats.md "bench.py --steps 5 --warmup 2 (compressed rows)" 7
head -14 $O/r_bench_kernel_stats.md | cut -c1-170
find $OUT -name "*.csv" -size +8M -delete

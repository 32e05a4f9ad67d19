#!/bin/bash
# zero-row skipping in the last layer's transposed aggregation: new tests, the NodeLoader options
# test, the bench line with and without it, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -m gpu -q -x \
  -k "rows_pack or skips_zero or training_split or fused_sage_stack" > $O/o_tests.log 2>&1
rc=$?; echo "new tests rc=$rc"; tail -15 $O/o_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout 300 python -m pytest tests/test_gpu_reference_install.py -m gpu -q -x -k "node_loader" > $O/o_tests2.log 2>&1
echo "loader test rc=$?"; tail -5 $O/o_tests2.log | cut -c1-300
for SG in 1 0; do
PYGAMD_SPARSE_GRAD=$SG timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/o_bench_sg$SG.json 2> $O/o_bench_sg$SG.err
echo "bench sparse_grad=$SG rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/o_bench_sg$SG.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('ms/step', round(d['ms_per_step'],2), r.get('kernel'), r.get('avg_launch_ms'), 'frac', r.get('frac'))
    print('others', r.get('others'))
    print('parity', d.get('parity_at_cpu_scale'))
except Exception as e:
    print('ERR', e)
PY
done
OUT=$O/prof_o; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stdout.log 2>&1)
echo "prof rc=$?"
python scripts/summarize_profile.py $(find $OUT -name "*kernel_stats.csv" | head -1) $O/o_bench_kernel_stats.md "bench.py --steps 5 --warmup 2 (zero-row skipping)" 7
head -22 $O/o_bench_kernel_stats.md | cut -c1-170
find $OUT -name "*.csv" -size +8M -delete

#!/bin/bash
# rows per wave of the sparse-source aggregation (R = 8): tests + bench + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -m gpu -q -x \
  -k "rows_pack or skips_zero or training_split" > $O/p_tests.log 2>&1
rc=$?; echo "new tests rc=$rc"; tail -5 $O/p_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
OUT=$O/prof_p; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stdout.log 2>&1)
echo "prof rc=$?"
python scripts/summarize_profile.py $(find $OUT -name "*kernel_stats.csv" | head -1) $O/p_bench_kernel_stats.md "bench.py --steps 5 --warmup 2" 7
grep "sparse\|rows_pack\|bits_count\|fused_fwd_kernel<long, 64\|Sum of" $O/p_bench_kernel_stats.md | cut -c1-200
find $OUT -name "*.csv" -size +8M -delete

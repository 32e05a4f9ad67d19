#!/bin/bash
# Round 3, final check on one MI355X: the whole GPU suite, smoke(), the default bench line (with
# the CPU baseline and the parity check at its scale), the uniform-graph variant.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > $O/z_tests.log 2>&1
echo "tests rc=$?"; tail -8 $O/z_tests.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke > $O/z_smoke.log 2>&1
echo "smoke rc=$?"; grep "\[smoke\]\|\[build\] torch" $O/z_smoke.log | cut -c1-400
timeout 600 python bench.py > $O/z_bench.json 2> $O/z_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/z_bench.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('ms/step', round(d['ms_per_step'],2), 'value', round(d['value']/1e9,3), 'G edges/s;', r.get('kernel'), r.get('avg_launch_ms'), 'frac', r.get('frac'), 'traffic', r.get('traffic'))
    print('others', r.get('others'), 'step', r.get('step'))
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'parity', d.get('parity_at_cpu_scale'))
except Exception as e:
    print('ERR', e)
PY
timeout 300 python bench.py --uniform --steps 20 --warmup 5 --no-cpu-baseline > $O/z_bench_uniform.json 2> $O/z_bench_uniform.err
echo "uniform rc=$?"; cut -c1-330 $O/z_bench_uniform.json
# kernel stats of the bench line as it stands at the end of the round
OUT=$O/prof_final; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stdout.log 2>&1)
echo "prof rc=$?"
python scripts/summarize_profile.py $(find $OUT -name "*kernel_stats.csv" | head -1) $O/z_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline (round 3, end of round; 7 steps, MI355X)" 7
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $O/z_bench_kernel_stats.csv
head -16 $O/z_bench_kernel_stats.md | cut -c1-170
find $OUT -name "*.csv" -size +8M -delete

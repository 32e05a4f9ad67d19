#!/bin/bash
# Round 3, first GPU session: parity of the new fused-kernel variants, then A/B timings of the
# products step (row-at-a-time vs streamed gather; two-launch vs one-launch input gradient).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_layers.py -x -q -m gpu > $O/a_tests1.log 2>&1
echo "tests1 rc=$?" > $O/a_status.txt
timeout 600 python -m pytest tests/test_gpu_scale.py::test_headline_step_parity_at_cpu_scale tests/test_gpu_ops.py -x -q -m gpu > $O/a_tests2.log 2>&1
echo "tests2 rc=$?" >> $O/a_status.txt
for cfg in "1 0" "2 0" "2 1" "1 1"; do
  set -- $cfg
  PYGAMD_FUSED_VARIANT=$1 PYGAMD_FUSE_BWD=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/a_bench_v$1_b$2.json 2> $O/a_bench_v$1_b$2.err
  echo "bench v$1 b$2 rc=$?" >> $O/a_status.txt
done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/a_bench_full.json 2> $O/a_bench_full.err
echo "bench full rc=$?" >> $O/a_status.txt
cat $O/a_status.txt
tail -3 $O/a_tests1.log $O/a_tests2.log
for f in $O/a_bench_v*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d['roofline']
    print(sys.argv[1], 'ms/step', round(d['ms_per_step'],2), 'dom', r.get('kernel'), r.get('avg_launch_ms'), 'others', r.get('others'), 'step', r.get('step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done

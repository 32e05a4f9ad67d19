#!/bin/bash
# Round 3, second GPU session: the producer/consumer fused layer (variants 3 / 4): parity tests,
# phase probe, A/B of the products step.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -k "sage_layer" -x -q -m gpu > $O/c_tests1.log 2>&1
echo "tests1 rc=$?" > $O/c_status.txt
timeout 400 python scripts/fused_probe.py --only-spec > $O/c_probe.log 2>&1
echo "probe rc=$?" >> $O/c_status.txt
for v in 3 4 1; do
  PYGAMD_FUSED_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/c_bench_v$v.json 2> $O/c_bench_v$v.err
  echo "bench v$v rc=$?" >> $O/c_status.txt
done
cat $O/c_status.txt
tail -5 $O/c_tests1.log
cat $O/c_probe.log
for f in $O/c_bench_v*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d['roofline']
    print(sys.argv[1], 'ms/step', round(d['ms_per_step'],2), 'dom', r.get('kernel'), r.get('avg_launch_ms'), 'others', r.get('others'), 'step', r.get('step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done

#!/bin/bash
# Producer/consumer fused layer: small launches in separate processes (stop at the first failure),
# then the parity tests, the phase probe and the products-step A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/d_debug.log
: > $O
for cfg in "3 3 256 1037" "3 0 256 1037" "4 0 256 1037" "3 0 256 40000" "3 0 100 40000" "4 0 100 40000"; do
  echo "== $cfg" >> $O
  timeout 45 python scripts/spec_debug.py $cfg >> $O 2>&1
  rc=$?
  echo "rc=$rc" >> $O
  if [ $rc -ne 0 ]; then grep -v "amdgpu.ids" $O | cut -c1-300; exit 1; fi
done
grep -v "amdgpu.ids" $O | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_gemm.py -k "sage_layer" -x -q -m gpu > gpurun_out/d_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/d_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 200 python scripts/fused_probe.py --only-spec > gpurun_out/d_probe.log 2>&1
rc=$?; echo "probe rc=$rc"; cat gpurun_out/d_probe.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in 3 4; do
  PYGAMD_FUSED_VARIANT=$v timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/d_bench_v$v.json 2> gpurun_out/d_bench_v$v.err
  echo "bench v$v rc=$?"
  python - gpurun_out/d_bench_v$v.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d['roofline']
    print(sys.argv[1], 'ms/step', round(d['ms_per_step'],2), 'dom', r.get('kernel'), r.get('avg_launch_ms'), 'others', r.get('others'), 'step', r.get('step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done

#!/bin/bash
# the opt-in split arithmetic on the end-of-round step (not the headline): bench line + default for reference
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for MODE in fp32 split; do
PYGAMD_GEMM_MODE=$MODE timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/u_bench_$MODE.json 2> $O/u_bench_$MODE.err
echo "bench gemm_mode=$MODE rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/u_bench_$MODE.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('ms/step', round(d['ms_per_step'],2), r.get('kernel'), r.get('avg_launch_ms'), 'others', r.get('others'))
    print(str(d['config'].get('gemm'))[:160])
except Exception as e:
    print('ERR', e)
PY
done

#!/bin/bash
# weight-gradient GEMMs on a side stream under the one-launch input gradient: correctness + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
PYGAMD_OVERLAP_WGRAD=1 timeout 300 python -m pytest tests/test_gpu_layers.py -m gpu -q -x \
  -k "fused_sage_stack or training_split" > $O/s_tests.log 2>&1
rc=$?; echo "tests (overlap on) rc=$rc"; tail -4 $O/s_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for CFG in "0 1" "1 1" "1 2" "0 1" "1 1"; do
set -- $CFG
PYGAMD_OVERLAP_WGRAD=$1 PYGAMD_OVERLAP_WGS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s_bench.json 2> $O/s_bench.err
echo "bench overlap=$1 wgs=$2 rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/s_bench.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('ms/step', round(d['ms_per_step'],2), r.get('kernel'), r.get('avg_launch_ms'), 'others', r.get('others'))
except Exception as e:
    print('ERR', e)
PY
done

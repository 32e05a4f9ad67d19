#!/bin/bash
# Round 3: compiled binding + RCCL-on-one-rank tests, the full suite, min/max probe, config timings
# on both binding routes, one products bench.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_binding.py -m gpu -x -q > $O/i_tests1.log 2>&1
rc=$?; echo "tests1 rc=$rc"; tail -6 $O/i_tests1.log
if [ $rc -ne 0 ]; then echo "(continuing)"; fi
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_binding.py --deselect tests/test_gpu_nccl.py --deselect tests/test_gpu_ops.py --deselect tests/test_gpu_gemm.py --deselect tests/test_gpu_layers.py --deselect tests/test_gpu_graph.py --deselect tests/test_gpu_configs.py --deselect tests/test_gpu_compile.py > $O/i_tests2.log 2>&1
rc=$?; echo "tests2 rc=$rc"; tail -4 $O/i_tests2.log
if [ $rc -ne 0 ]; then echo "(continuing)"; fi
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_r03_minmax -o minmax -- python $GRAFT_REPO_ROOT/scripts/reduce_probe.py > $O/i_reduce_probe.log 2>&1
echo "reduce probe rc=$?"; grep "fused " $O/i_reduce_probe.log
cd $GRAFT_REPO_ROOT
SKIP_GRAPH=1 timeout 200 python scripts/time_configs.py > $O/i_configs_compiled.log 2>&1; grep "config" $O/i_configs_compiled.log
SKIP_GRAPH=1 PYGAMD_BINDING=ctypes timeout 200 python scripts/time_configs.py > $O/i_configs_ctypes.log 2>&1; grep "config" $O/i_configs_ctypes.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/i_bench.json 2> $O/i_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/i_bench.json').read().strip().splitlines()[-1])
    print('ms/step', round(d['ms_per_step'],2), d['roofline'].get('kernel'), d['roofline'].get('avg_launch_ms'), d['roofline'].get('others'))
except Exception as e:
    print('ERR', e)
PY

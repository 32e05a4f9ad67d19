#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "minmax or inplace or out_of_range or scatter_errors" > gpurun_out/h_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/h_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_minmax -o minmax -- python $GRAFT_REPO_ROOT/scripts/reduce_probe.py > $GRAFT_REPO_ROOT/gpurun_out/h_reduce_probe.log 2>&1
echo "reduce probe rc=$?"; grep -v amdgpu.ids $GRAFT_REPO_ROOT/gpurun_out/h_reduce_probe.log | tail -8
ls $GRAFT_REPO_ROOT/gpurun_out/prof_r03_minmax/*/ 2>/dev/null | head

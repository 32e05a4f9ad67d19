#!/bin/bash
# Round 3: full GPU suite after the ADVICE fixes / sampler options / atomic-free min-max backward,
# then the reduction probe under the kernel tracer.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/g_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_minmax -o minmax -- python $GRAFT_REPO_ROOT/scripts/reduce_probe.py > $GRAFT_REPO_ROOT/gpurun_out/g_reduce_probe.log 2>&1
echo "reduce probe rc=$?"; grep -v amdgpu.ids $GRAFT_REPO_ROOT/gpurun_out/g_reduce_probe.log | tail -8

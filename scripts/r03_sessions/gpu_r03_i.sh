#!/bin/bash
# Round 3: static-shape mini-batch path — tests, then config 4 at the full papers100M shape, eager
# vs captured.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_minibatch_capture.py tests/test_gpu_sampler.py -m gpu -x -q > $O/j_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -25 $O/j_tests.log | cut -c1-220
if [ $rc -ne 0 ]; then exit 1; fi
timeout 500 python bench.py --mode minibatch --capture --steps 200 --warmup 20 > $O/j_mb_capture.json 2> $O/j_mb_capture.err
echo "capture rc=$?"; tail -3 $O/j_mb_capture.err | cut -c1-300; cut -c1-1600 $O/j_mb_capture.json
timeout 500 python bench.py --mode minibatch --steps 200 --warmup 20 > $O/j_mb_eager.json 2> $O/j_mb_eager.err
echo "eager rc=$?"; cut -c1-400 $O/j_mb_eager.json

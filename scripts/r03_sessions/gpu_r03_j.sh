#!/bin/bash
cd /tmp
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
PYGAMD_CAPTURE=0 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r03_mb_capture -o mb -- python $GRAFT_REPO_ROOT/bench.py --mode minibatch --capture --steps 30 --warmup 5 > $O/k_mb_capture_prof.json 2> $O/k_mb_capture_prof.err
echo "rc=$?"; cut -c1-200 $O/k_mb_capture_prof.json

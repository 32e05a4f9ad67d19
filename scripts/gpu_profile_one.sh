#!/bin/bash
# rocprofv3 kernel stats of one script: bash scripts/gpu_profile_one.sh <name> <script.py> [args]
set -u
export TMPDIR=/tmp
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python $GRAFT_REPO_ROOT/"$@" > $OUT/stdout.log 2>&1)
grep -v "amdgpu.ids\|W2026\|E2026" $OUT/stdout.log | tail -2
find $OUT -name "*kernel_trace*.csv" -size +8M -delete

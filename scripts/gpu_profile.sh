#!/bin/bash
# rocprofv3 kernel trace of the default bench command; summary goes to gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}
shift || true
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_stdout.log 2>&1
F=$(find $OUT -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && cut -c1-110 "$F" | head -24
tail -2 $OUT/bench_stdout.log
# keep the merge small: drop the raw per-dispatch trace if it is huge
find $OUT -name "*kernel_trace*.csv" -size +20M -delete

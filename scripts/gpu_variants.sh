#!/bin/bash
export TMPDIR=/tmp
run() { echo "== $*"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],2), 'ms/step', round(d['value']/1e9,3), 'G edges/s; F256 launch', r['avg_launch_ms'], 'ms', r['achieved'], 'GB/s frac', r['frac'], r['per_width_avg_ms'])"; }
run
run --index-dtype int32
run --uniform
run --uniform --index-dtype int32

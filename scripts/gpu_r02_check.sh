#!/bin/bash
# round-2 check on the GPU box: GPU test suite (with durations), then the default bench line.
# Usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r02_check.sh <tag> [pytest args]'
tag=${1:-r02}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname > gpurun_out/${tag}_device.txt 2>&1; nproc >> gpurun_out/${tag}_device.txt; free -g >> gpurun_out/${tag}_device.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=20 "$@" > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -40 gpurun_out/${tag}_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cat gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err

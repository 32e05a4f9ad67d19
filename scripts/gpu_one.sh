#!/bin/bash
# run a subset of GPU tests: bash scripts/gpu_one.sh <pytest args>
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest "$@" -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -40 | tee gpurun_out/pytest_one.log

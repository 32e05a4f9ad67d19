#!/bin/bash
# run one python script on the GPU box: bash scripts/gpu_run_py.sh <script.py> [args]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python "$@" 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/run_py.log

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_configs.py tests/test_capi.py -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -30 | tee gpurun_out/pytest_one.log
timeout 600 python scripts/time_configs.py 2>&1 | tail -6 | tee gpurun_out/time_configs.log

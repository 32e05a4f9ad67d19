#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_layers.py -k "segment_matmul or rgcn or RGCN or hetero" -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -8
timeout 300 python scripts/time_rgcn.py 2>&1 | tail -1

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_ops.py -k "gat or GAT or head_dot or softmax" -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -8
timeout 300 python scripts/time_configs.py 2>&1 | grep config3

#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py tests/test_gpu_scale.py tests/test_capi.py -k "max or min or capi or header or symbol or sage or graph_conv or aggregation" -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -12
timeout 300 python scripts/reduce_probe.py 2>&1 | grep fused

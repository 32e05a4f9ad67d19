#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py tests/test_gpu_sampler.py -k "relu_backward or sage or Sage or fused or stack or hop" -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -6
echo "== relu epilogue OFF: tests"
PYGAMD_RELU_EPILOGUE=0 timeout 900 python -m pytest tests/test_gpu_layers.py -k "sage or Sage or fused or stack" -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -6
echo "== bench (default)"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
echo "== bench (epilogue off)"
PYGAMD_RELU_EPILOGUE=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330

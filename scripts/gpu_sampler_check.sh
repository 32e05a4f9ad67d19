#!/bin/bash
# sampler parity tests + config-4 timing
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_capi.py -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -30 | tee gpurun_out/pytest_sampler.log
timeout 600 python scripts/time_minibatch.py 2>&1 | tail -6 | tee gpurun_out/time_minibatch.log
timeout 600 python bench.py --mode minibatch --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_minibatch.log

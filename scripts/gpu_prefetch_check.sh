#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sampler.py -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -12
for P in 0 2; do
  timeout 600 python bench.py --mode minibatch --steps 60 --warmup 10 --no-cpu-baseline --prefetch $P 2>&1 | tail -1 | cut -c1-260
done

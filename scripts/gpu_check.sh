#!/bin/bash
# One GPU-box visit: parity tests, smoke, a scaled-down and a full-size bench line.
# Usage (via gpurun): bash scripts/gpu_check.sh [quick|full]
set -u
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1 TMPDIR=/tmp
MODE=${1:-full}
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; free -g | head -2 >> gpurun_out/device.txt
echo "== pytest -m gpu" 
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -200 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench 1/16"
timeout 600 python bench.py --scale 0.0625 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/bench_s16.log
if [ "$MODE" = "full" ]; then
  echo "== bench full"
  timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench_full.log
fi

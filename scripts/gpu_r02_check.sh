#!/bin/bash
# round-2 check on the GPU box: GPU test suite (with durations), smoke, then the bench line with
# the repo's own GEMM kernels and (for comparison) with the library GEMMs.
# Usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r02_check.sh <tag> [pytest args]'
tag=${1:-r02}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname > gpurun_out/${tag}_device.txt 2>&1; nproc >> gpurun_out/${tag}_device.txt; free -g >> gpurun_out/${tag}_device.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=12 "$@" > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/${tag}_tests.log | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cut -c1-1500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
PYGAMD_GEMM=lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_lib.json 2>> gpurun_out/${tag}_bench.err
echo "bench(lib) rc=$?"; cut -c1-400 gpurun_out/${tag}_bench_lib.json
PYGAMD_GEMM_MODE=split timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_split.json 2>> gpurun_out/${tag}_bench.err
echo "bench(split mode) rc=$?"; cut -c1-400 gpurun_out/${tag}_bench_split.json

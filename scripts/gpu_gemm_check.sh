#!/bin/bash
# GEMM parity tests + timing probe (+ optional rocprof) on the GPU box.
tag=${1:-gemm}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -40 | tee gpurun_out/${tag}_tests.log
timeout 600 python scripts/gemm_probe.py 2>&1 | tee gpurun_out/${tag}_probe.log

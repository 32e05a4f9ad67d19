#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 200 python scripts/fused_probe.py --prio 2>&1 | grep -v amdgpu.ids
timeout 120 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k softmax 2>&1 | tail -3

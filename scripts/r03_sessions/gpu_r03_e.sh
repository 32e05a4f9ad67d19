#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 45 python scripts/spec_debug.py 3 0 256 40000 2>&1 | grep -v amdgpu.ids
if [ ${PIPESTATUS[0]} -ne 0 ]; then exit 1; fi
timeout 200 python scripts/fused_probe.py --only-spec --widths 256 > gpurun_out/f_probe.log 2>&1
rc=$?; echo "probe rc=$rc"; cat gpurun_out/f_probe.log

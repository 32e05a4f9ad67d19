#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 scripts/probes/hwid_probe.hip -o /tmp/hwid 2>/dev/null && timeout 30 /tmp/hwid > gpurun_out/e_hwid.log 2>&1
cat gpurun_out/e_hwid.log
timeout 45 python scripts/spec_debug.py 3 0 256 40000 2>&1 | grep -v amdgpu.ids
if [ ${PIPESTATUS[0]} -ne 0 ]; then exit 1; fi
timeout 45 python scripts/spec_debug.py 4 0 100 40000 2>&1 | grep -v amdgpu.ids
if [ ${PIPESTATUS[0]} -ne 0 ]; then exit 1; fi
timeout 200 python scripts/fused_probe.py --only-spec > gpurun_out/e_probe.log 2>&1
rc=$?; echo "probe rc=$rc"; cat gpurun_out/e_probe.log

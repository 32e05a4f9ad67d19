#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python scripts/gather_lines_probe.py 2>&1 | grep -v amdgpu.ids

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 python scripts/smoke_debug.py 1024 1 31 2>&1 | grep -v amdgpu.ids
timeout 200 python __graft_entry__.py smoke 2>&1 | grep "smoke\|Error\|assert" | cut -c1-400

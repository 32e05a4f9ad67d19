#!/bin/bash
# Final round-2 measurements in one GPU-box visit (summaries are copied to profiles/ by
# scripts/collect_r02_profiles.py <tag>):
#   rocprofv3 --kernel-trace --stats of the default bench line; the two PMC passes (FETCH_SIZE,
#   WRITE_SIZE: separate passes, kernel-trace only); kernel stats of the split-mode bench line;
#   the informational config timings.
set -u
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
source <(sed -n '/^prof() {/,/^}/p;/^pmc() {/,/^}/p' $R/scripts/gpu_r02_profile.sh)
echo "== kernel stats: bench"
prof bench python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
echo "== PMC FETCH_SIZE"
pmc fetch FETCH_SIZE python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
echo "== PMC WRITE_SIZE"
pmc write WRITE_SIZE python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
echo "== kernel stats: bench, PYGAMD_GEMM_MODE=split"
PYGAMD_GEMM_MODE=split prof bench_split python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
echo "== configs"
timeout 200 python $R/scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/${TAG}_configs.txt | tail -6

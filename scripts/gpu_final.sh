#!/bin/bash
# Round-end validation in one GPU-box visit: full parity suite, smoke, the bench line, and the
# rocprofv3 kernel stats of the bench / the secondary configs / the mini-batch mode.
set -u
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1 TMPDIR=/tmp
TAG=${1:-final}
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider --tb=short --no-header 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench full"
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_full.log | cut -c1-200
prof() {  # prof <name> <cmd...>
  local OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$1; shift
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- "$@" > $OUT/stdout.log 2>&1)
  grep -v "W2026\|E2026" $OUT/stdout.log | tail -4 | cut -c1-300
  find $OUT -name "*kernel_trace*.csv" -size +8M -delete
}
echo "== profile bench"
prof bench python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline
echo "== profile configs"
prof configs python $GRAFT_REPO_ROOT/scripts/time_configs.py
echo "== profile minibatch"
prof minibatch python $GRAFT_REPO_ROOT/bench.py --mode minibatch --steps 30 --warmup 5 --no-cpu-baseline

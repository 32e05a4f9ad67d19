#!/bin/bash
# Round 6, after slots.SlotTrainer: the whole GPU suite, the default bench line, the mini-batch
# profiles (captured + eager kernel stats) and the eager step's kernel-by-kernel timeline
# (PYGAMD_SLOT_PIPELINE=0 there: one stream, so that seed-to-seed intervals are whole steps).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1; tail -3 gpurun_out/full_gpu.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
bash scripts/gpu_r06_profile.sh
OUT=$R/gpurun_out/mbtrace_trainer; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && PYGAMD_CAPTURE=0 PYGAMD_SLOT_PIPELINE=0 timeout 600 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/bench.py --mode minibatch --capture --steps 6 --warmup 2 --scale 0.25 > $OUT/stdout.log 2>&1)
python scripts/minibatch_step_trace.py $(find $OUT -name "*kernel_trace.csv" | head -1) > gpurun_out/r06_minibatch_step_timeline_trainer.txt 2>&1
tail -30 gpurun_out/r06_minibatch_step_timeline_trainer.txt
find $OUT -name "*.csv" -size +4M -delete

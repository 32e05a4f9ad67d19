#!/bin/bash
# Round 6: slots.SlotTrainer — its tests, then the captured config-4 step A/B on the same box
# (PYGAMD_SLOT_TRAINER=0 = the round-5 step through autograd + ATen loss + torch's fused Adam;
# PYGAMD_SLOT_PIPELINE=0 = the trainer without the parallel sampling branch).
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_slots.py tests/test_gpu_minibatch_capture.py -m gpu -x -q > gpurun_out/trainer_tests.log 2>&1
tail -5 gpurun_out/trainer_tests.log
run() {
  PYGAMD_SLOT_TRAINER=$1 PYGAMD_SLOT_PIPELINE=$2 python bench.py --mode minibatch --capture --steps 300 --warmup 30 > gpurun_out/mb_trainer_$1$2.json 2> gpurun_out/mb_trainer_$1$2.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/mb_trainer_$1$2.json').read().strip().splitlines()[-1])
    print('trainer=$1 pipeline=$2', round(d['ms_per_step'], 4), 'ms/step')
except Exception as e:
    print('trainer=$1 pipeline=$2 FAILED', e)
    print(open('gpurun_out/mb_trainer_$1$2.err').read()[-2000:])
PY
}
run 1 1; run 1 0; run 0 0; run 1 1; run 1 0

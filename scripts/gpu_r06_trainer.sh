#!/bin/bash
# Round 6: slots.SlotTrainer — its tests, then the captured config-4 step A/B on the same box
# (PYGAMD_SLOT_TRAINER=0 = the round-5 step through autograd + ATen loss + torch's fused Adam).
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_slots.py tests/test_gpu_minibatch_capture.py -m gpu -x -q > gpurun_out/trainer_tests.log 2>&1
tail -5 gpurun_out/trainer_tests.log
for t in 1 0 1 0; do
  PYGAMD_SLOT_TRAINER=$t python bench.py --mode minibatch --capture --steps 300 --warmup 30 > gpurun_out/mb_trainer_$t.json 2> gpurun_out/mb_trainer_$t.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/mb_trainer_$t.json').read().strip().splitlines()[-1])
print('trainer=$t', round(d['ms_per_step'], 4), 'ms/step')
PY
done

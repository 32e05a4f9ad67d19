#!/bin/bash
# exercise the torch.distributed (RCCL) code path of bench.py with a world of 1 rank, both modes
export TMPDIR=/tmp
for MODE in fullbatch minibatch; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29511 bench.py --gpus 2 --mode $MODE --steps 5 --warmup 2 --scale 0.0625 \
    --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | grep -E "^\{|Error|error" | tail -2 | cut -c1-330
done

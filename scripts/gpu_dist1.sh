#!/bin/bash
# exercise the torch.distributed (RCCL) code path of bench.py with a world of 1 rank
export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --scale 0.0625 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-260

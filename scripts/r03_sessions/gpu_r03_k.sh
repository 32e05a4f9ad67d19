#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_minibatch_capture.py -m gpu -x -q > $O/l_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/l_tests.log | cut -c1-200
if [ $rc -ne 0 ]; then exit 1; fi
timeout 500 python bench.py --mode minibatch --capture --steps 200 --warmup 20 > $O/l_mb_capture.json 2> $O/l_mb_capture.err
echo "capture rc=$?"; cut -c1-330 $O/l_mb_capture.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --init-dist --mode minibatch --capture --steps 100 --warmup 10 > $O/l_mb_capture_dist.json 2> $O/l_mb_capture_dist.err
echo "capture+dist rc=$?"; grep "^{" $O/l_mb_capture_dist.json | cut -c1-330
timeout 400 python -m pytest tests/test_gpu_nccl.py -m gpu -x -q > $O/l_nccl.log 2>&1
echo "nccl tests rc=$?"; tail -3 $O/l_nccl.log | cut -c1-300

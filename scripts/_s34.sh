mkdir -p gpurun_out/s34
timeout 600 python -m pytest tests -m gpu -q -x -k "colsum or head_dot or gat or GAT or bias or relu or layers or config" > gpurun_out/s34/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/s34/pytest.log
tail -3 gpurun_out/s34/pytest.log
ONLY=configs bash scripts/gpu_r05_profile.sh

#!/bin/bash
# TunableOp pass over the secondary configs (GCN/Cora, GAT/arxiv, RGCN/FB15k shapes), then a
# re-run reading products table + the new entries.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
CSV=$GRAFT_REPO_ROOT/gpurun_out/tunableop_configs.csv
rm -f $GRAFT_REPO_ROOT/gpurun_out/tunableop_configs*.csv
SKIP_GRAPH=1 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$CSV \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 \
  timeout 900 python scripts/time_configs.py 2>&1 | grep config
F=$(ls gpurun_out/tunableop_configs*.csv | head -1)
wc -l $F
python - <<PY
base = open('pytorch_geometric_amd/tuning/gemm_mi355x_products.csv').read().splitlines()
new = open('$F').read().splitlines()
have = {l.split(',')[1] for l in base if not l.startswith('Validator')}
add = [l for l in new if not l.startswith('Validator') and l.split(',')[1] not in have]
open('gpurun_out/gemm_merged.csv', 'w').write('\n'.join(base + add) + '\n')
print('merged', len(base), '+', len(add))
PY
TUNED_GEMM=$GRAFT_REPO_ROOT/gpurun_out/gemm_merged.csv timeout 600 python scripts/time_configs.py 2>&1 | grep -E "config|tuned"

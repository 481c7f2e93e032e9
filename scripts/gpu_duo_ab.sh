#!/bin/bash
# A/B on one box, interleaved: the default GEMM heuristic (gemm_duo_kernel for QKV / out-projection) vs MDR_GEMM_CFG=60 (without it)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_encoder_gpu.py -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for cfg in 0 60; do
    echo "-- MDR_GEMM_CFG=$cfg round $r"
    MDR_GEMM_CFG=$cfg timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 30 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'hop2_encode', r['stage_ms']['hop2_encode'], 'seq', r['sequential']['value'], r['sequential']['stage_ms']['hop1_encode'], r['sequential']['stage_ms']['hop2_encode'])"
  done
done

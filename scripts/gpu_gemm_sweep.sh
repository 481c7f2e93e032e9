#!/bin/bash
python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -2
MDR_GEMM_CFG=4 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -2
for GN in 0 1 2 4 8; do
  MDR_GEMM_GN=$GN python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rows 1000000 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('gn', $GN, 'ms/step', r['ms_per_step'], 'hop1', r['stage_ms']['hop1_encode'], 'hop2', r['stage_ms']['hop2_encode'])"
done
bash scripts/gpu_prof_bench.sh profb6 --rows 1000000 2>&1 | grep -E "hop1_encode|  .*gemm_persist|  .*attention_kernelILi24|  .*layernorm" | head -8

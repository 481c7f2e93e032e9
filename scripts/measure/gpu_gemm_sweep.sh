#!/bin/bash
MDR_GEMM_WIDE=2 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -1
for W in 0 2; do
  MDR_GEMM_WIDE=$W python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rows 1000000 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('wide', $W, 'hop1', r['stage_ms']['hop1_encode'], 'hop2', r['stage_ms']['hop2_encode'])"
done
MDR_GEMM_WIDE=2 bash scripts/measure/gpu_prof_bench.sh profb8 --rows 1000000 2>&1 | grep -E "   .*gemm_persist|   .*attention_kernelILi24|   .*layernorm" | head -8

#!/bin/bash
for CFG in 0 1 2 3; do
  MDR_GEMM_CFG=$CFG python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rows 1000000 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cfg', $CFG, 'ms/step', r['ms_per_step'], 'hop1', r['stage_ms']['hop1_encode'], 'hop2', r['stage_ms']['hop2_encode'])"
done
for CFG in 0 1 2 3; do MDR_GEMM_CFG=$CFG python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -1; done

#!/bin/bash
# fp16 vs fp32 residual stream (mdr_encoder_config.residual_fp32): embedding error, retrieval agreement, step time.
TAG=${1:-resid}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for m in 0 1; do
  echo "=== MDR_RESIDUAL_FP32=$m"
  MDR_RESIDUAL_FP32=$m timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_retrieval_agreement_gpu.py -m gpu -q -s 2>&1 | grep -E "encoder (tiny|base)|retrieval agreement|passed|failed|large-batch" | cut -c1-900
  MDR_RESIDUAL_FP32=$m timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_resid$m.json 2> $OUT/bench_resid$m.err
  python - <<PY
import json
r=json.load(open("$OUT/bench_resid$m.json"))
print("bench residual_fp32=$m:", r["value"], "q/s", r["ms_per_step"], "ms", r["stage_ms"], r["self_check"]["full_size_exact"])
PY
done 2>&1 | tee $OUT/summary.txt

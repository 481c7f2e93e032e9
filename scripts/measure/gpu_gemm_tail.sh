#!/bin/bash
# round 4: GEMM tail tiles -- parity + per-step distribution + GEMM sweep over M
set -u
TAG=${1:-r04h}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "== GEMM time over M (kernel 0 = the forward's choice)"
timeout 300 python scripts/measure/gpu_gemm_msweep.py 2>&1 | tail -30
echo "== bench (pipelined loop, pool 16)"
timeout 900 python bench.py --no-cpu-baseline --no-anisotropic > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
    print("per_step", json.dumps(r.get("per_step"))[:700])
    print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"])
    print("numerics", {k: v["value"] for k, v in r.get("numerics_modes", {}).items()})
    print("self_check", r["self_check"]["full_size_exact"])
except Exception as e:
    print("bench parse failed", e)
PY

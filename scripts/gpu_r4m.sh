#!/bin/bash
# round 4: attention kernel experiment -- parity + kernel time (rocprof stats of a short bench)
set -u
TAG=${1:-r04m}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_retrieval_agreement_gpu.py -m gpu -x -q 2>&1 | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o b -- python $REPO/bench.py --no-cpu-baseline --no-anisotropic --no-sequential --no-verify --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
S=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && grep -E "attention|Name" "$S" | cut -c1-160
rm -rf $OUT/p
python - $OUT/bench.json <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
except Exception as e:
    print("parse failed", e)
PY

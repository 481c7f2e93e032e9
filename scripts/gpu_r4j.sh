#!/bin/bash
# round 4: hop-1 group sweep of the pipelined bench loop
set -u
TAG=${1:-r04j}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for G in 1 2 4 8; do
  timeout 600 python bench.py --hop1-group $G --steps 24 --warmup 9 --no-cpu-baseline --no-anisotropic --no-sequential --no-verify > $OUT/bench_g$G.json 2> $OUT/bench_g$G.err
  python - $OUT/bench_g$G.json $G <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("G", sys.argv[2], "value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"], "per_step ms", r["per_step"]["ms"])
except Exception as e:
    print("parse failed", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
done

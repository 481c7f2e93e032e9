#!/bin/bash
# round 4: CLI bench + default bench with the deep loop
set -u
TAG=${1:-r04e}; shift || true
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_r4b.sh $TAG
echo "== bench DEFAULT (deep loop)"
timeout 900 python bench.py --no-cpu-baseline --no-anisotropic > $OUT/bench_deep.json 2> $OUT/bench_deep.err; tail -3 $OUT/bench_deep.err
python - $OUT/bench_deep.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
    print("per_step", json.dumps(r.get("per_step"))[:600])
    print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms")})
    print("one_deep", r.get("pipelined_one_deep", {}).get("value"), r.get("pipelined_one_deep", {}).get("stage_ms"))
    print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"])
    print("numerics", {k: v["value"] for k, v in r.get("numerics_modes", {}).items()})
    print("self_check", r["self_check"])
except Exception as e:
    print("bench parse failed", e)
PY

#!/bin/bash
# Round-3 MIPS visit: the i8 / fullsize / mips parity tests with printed lines, then the default bench. -> gpurun_out/<tag>/
set -u
TAG=${1:-r03m}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest mips"
timeout 1500 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_fullsize_gpu.py tests/test_mips_gpu.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_mips.txt
grep -E "int8 tier on aniso|screen telemetry|passed|failed|Error|assert" $OUT/pytest_mips.txt | cut -c1-900 | tail -30
tail -2 $OUT/pytest_mips.txt
echo "== bench default"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("aniso", json.dumps(r.get("anisotropic")))
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
PY

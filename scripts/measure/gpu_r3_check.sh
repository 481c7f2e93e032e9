#!/bin/bash
# Round-3 visit: GPU parity suite (with the printed distance lines), smoke, the default bench line. -> gpurun_out/<tag>/
set -u
TAG=${1:-r03a}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import torch; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_gpu_full.txt
grep -E "^encoder |retrieval agreement|sharded selftest|ids==one-index|passed|failed|Error|error" $OUT/pytest_gpu_full.txt | cut -c1-600 | tail -60
tail -3 $OUT/pytest_gpu_full.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench default"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-1500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("enc", {k: r["roofline_encoder"][k] for k in ("achieved", "frac")}, r["roofline_encoder"]["hop2"])
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("cpu", r["cpu_baseline"])
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
PY

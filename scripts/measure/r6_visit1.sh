#!/bin/bash
# round 6, first visit: full GPU suite with the new tests, structured-corpus MIPS sub-results (short passages), query-group A/B, default bench line
set -u
TAG=${1:-r06v1}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import torch, os; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))" > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
grep -o "clustered 5 M rows.*" $OUT/pytest_gpu_full.txt | cut -c1-300
echo "== structured (short passages)"
timeout 900 python bench.py --mode structured > $OUT/structured_short.json 2> $OUT/structured_short.err; tail -2 $OUT/structured_short.err; cut -c1-3000 $OUT/structured_short.json
echo "== query groups A/B"
for M in 0 1 2; do
  MDR_MIPS_EVEN_GROUPS=$M timeout 300 python scripts/measure/r6_groups_ab.py 5000000 f32x2h 260:8 300:8 400:8 300:1 800:1 800:8 2>&1 | grep groups= | tee -a $OUT/groups.txt
  MDR_MIPS_EVEN_GROUPS=$M timeout 300 python scripts/measure/r6_groups_ab.py 6250000 bf16 300:8 800:8 800:100 2>&1 | grep groups= | tee -a $OUT/groups.txt
  MDR_MIPS_EVEN_GROUPS=$M timeout 300 python scripts/measure/r6_groups_ab.py 625000 f32x2h 800:1 400:4 2>&1 | grep groups= | tee -a $OUT/groups.txt
done
echo "== bench DEFAULT"
/usr/bin/time -v timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; grep "Elapsed (wall" $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
print("structured", json.dumps(r.get("structured"))[:3000])
PY
du -sh $OUT

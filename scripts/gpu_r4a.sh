#!/bin/bash
# round 4, first GPU visit: env probe, parity tests (new: multi-rank CLI, pipeline variants), default bench (pool of batches, numerics modes), bench --mode cli
set -u
TAG=${1:-r04a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(python -c "import torch, os; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
 free -g | head -2; df -h /dev/shm /tmp | cat; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null) > $OUT/env.txt 2>&1
cat $OUT/env.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_gpu_full.txt; tail -15 $OUT/pytest_gpu_full.txt
grep -o "encoder [a-z0-9]*\.[a-z]* residual_fp32=[012]: e_regime.*" $OUT/pytest_gpu_full.txt > $OUT/encoder_distances.txt
echo "== bench DEFAULT"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
    print("per_step", json.dumps(r.get("per_step")))
    print("numerics", json.dumps(r.get("numerics_mode")), json.dumps(r.get("numerics_modes")))
    print("enc", {k: r["roofline_encoder"][k] for k in ("achieved", "frac")})
    print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"])
    print("self_check", r["self_check"]["full_size_exact"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "== bench --mode cli (5M)"
timeout 1200 python bench.py --mode cli > $OUT/bench_cli.json 2> $OUT/bench_cli.err; tail -25 $OUT/bench_cli.err | cut -c1-300; cut -c1-3000 $OUT/bench_cli.json
du -sh $OUT

#!/bin/bash
# One GPU-box visit: parity tests, smoke, the default bench line, MIPS-only benches, rocprofv3 kernel stats. Everything lands in gpurun_out/<tag>/.
# usage: scripts/measure/gpu_round.sh <tag> [extra bench args]     (counter passes: scripts/measure/gpu_pmc_screen.sh)
set -u
TAG=${1:-r03}; shift || true
EXTRA="$@"
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.txt 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/env.txt

echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
(grep -o "encoder [a-z0-9]*\.[a-z]* residual_fp32=[01]: e_regime.*" $OUT/pytest_gpu_full.txt; grep -o "retrieval agreement, residual_fp32=[A-Za-z]*.*" $OUT/pytest_gpu_full.txt;
 grep -o "int8 tier on anisotropic rows.*" $OUT/pytest_gpu_full.txt; grep "ids==one-index\|sharded selftest" $OUT/pytest_gpu_full.txt; tail -1 $OUT/pytest_gpu_full.txt) > $OUT/pytest_gpu.txt

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt

echo "== bench 1M (MIPS only)"
timeout 600 python bench.py --rows 1000000 --steps 50 --warmup 5 --no-encoder --no-cpu-baseline $EXTRA > $OUT/bench_1m_mips.json 2> $OUT/bench_1m_mips.err; cut -c1-300 $OUT/bench_1m_mips.json
echo "== bench 5M (MIPS only)"
timeout 900 python bench.py --rows 5000000 --steps 20 --warmup 3 --no-encoder --no-cpu-baseline $EXTRA > $OUT/bench_5m_mips.json 2> $OUT/bench_5m_mips.err; cut -c1-300 $OUT/bench_5m_mips.json
echo "== bench DEFAULT (5M, 2-hop with encoder, cpu baseline, anisotropic sub-result) -- the headline line"
timeout 900 python bench.py $EXTRA > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("enc", {k: r["roofline_encoder"][k] for k in ("achieved", "frac", "frac_of_sustained_mfma")})
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["top1_id_agreement_with_hip_index"])
print("aniso", json.dumps(r.get("anisotropic"))[:600])
print("residual_fp32", r.get("residual_fp32"))
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
PY

echo "== rocprofv3 kernel stats (default bench command, no cpu baseline)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $REPO/bench.py --no-cpu-baseline $EXTRA > $OUT/prof_stats.log 2>&1
S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/kernel_stats.csv; cut -c1-150 "$S" | head -14)
rm -rf $OUT/prof_stats
find $OUT -name "*.db" -size +20M -delete 2>/dev/null
du -sh $OUT

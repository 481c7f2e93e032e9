#!/bin/bash
# round 6: the closing visit -- parity tests, smoke, default bench line, MIPS-only benches, CLI bench, rocprof kernel stats, FETCH_SIZE pass for pmc_traffic.json
set -u
TAG=${1:-r06final}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import torch, os; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))" > $OUT/env.txt 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/env.txt
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
(grep -o "encoder [a-z0-9]*\.[a-z]* residual_fp32=[012]: e_regime.*" $OUT/pytest_gpu_full.txt; grep -o "   literal-O1 regime error.*" $OUT/pytest_gpu_full.txt; grep -o "retrieval agreement, residual_fp32=[A-Za-z0-9]*.*" $OUT/pytest_gpu_full.txt;
 grep -o "screen-k at 5 M rows.*" $OUT/pytest_gpu_full.txt; grep "ids==one-index\|sharded selftest" $OUT/pytest_gpu_full.txt; tail -1 $OUT/pytest_gpu_full.txt) > $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
echo "== FETCH_SIZE pass (pmc_traffic.json)"
PMC_FETCH_ONLY=1 PMC_PROFILE_TAG=r06_mips5m_pmc bash scripts/measure/gpu_pmc_screen.sh $TAG/pmc > $OUT/pmc.log 2>&1; grep "pmc_traffic.json:" $OUT/pmc.log | cut -c1-200
[ -f $OUT/pmc/pmc_traffic.json ] && cp $OUT/pmc/pmc_traffic.json $REPO/profiles/pmc_traffic.json
echo "== bench 5M (MIPS only)"
timeout 900 python bench.py --rows 5000000 --steps 20 --warmup 3 --no-encoder --no-cpu-baseline > $OUT/bench_5m_mips.json 2> $OUT/bench_5m_mips.err; cut -c1-200 $OUT/bench_5m_mips.json
echo "== bench DEFAULT -- the headline line"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("enc", {k: r["roofline_encoder"][k] for k in ("achieved", "frac")})
print("per_step", json.dumps(r.get("per_step"))[:500])
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["top1_id_agreement_with_hip_index"])
print("numerics", r.get("numerics_mode"), {k: v["value"] for k, v in r.get("numerics_modes", {}).items()})
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
for name, v in (r.get("structured") or {}).items():
    for nq in ("nq100", "nq200"):
        x = v[nq]
        print(f"structured {name:17s} {nq}: {x['ms_per_search']:.4f} ms  int8 decided {x['int8_tier_decided']}  emitted {x['candidates_emitted']:7d} rescored {x['candidates_rescored']:6d}  agree {x['top1_agreement_up_to_exact_ties']}")
PY
echo "== bench --mode cli"
timeout 1200 python bench.py --mode cli > $OUT/bench_cli.json 2> $OUT/bench_cli.err; python - $OUT/bench_cli.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("device_loop", r.get("device_loop", {}).get("value"))
for k in ("cli_default", "cli_device"):
    print(k, {x: r[k][x] for x in ("value", "ms_per_batch", "steady_state_queries_per_s", "whole_process_seconds")}, r[k].get("startup_s"))
print("ratio", r.get("cli_over_device_loop"), "identical", r.get("legs_jsonl_identical"))
PY
echo "== bench --mode cli, two ranks on this one GPU over gloo, started by bench.py itself (the N > 1 door of the CLI leg)"
timeout 900 python bench.py --mode cli --gpus 2 --share-gpu --backend gloo --rows 400000 --questions 1000 --no-sequential --cli-legs default > $OUT/bench_cli_2ranks_shared.json 2> $OUT/bench_cli_2ranks_shared.err
python - $OUT/bench_cli_2ranks_shared.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("n_gpus", r["n_gpus"], {x: r["cli_default"][x] for x in ("value", "records", "whole_process_seconds")})
except Exception as e:
    print("2-rank cli leg failed:", e)
PY
echo "== rocprofv3 kernel stats (default bench command, no cpu baseline)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-anisotropic --structured > $OUT/prof_stats.log 2>&1
S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/kernel_stats.csv; cut -c1-150 "$S" | head -12)
rm -rf $OUT/prof_stats $OUT/pmc/p
find $OUT -name "*.db" -size +20M -delete 2>/dev/null
cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
du -sh $OUT

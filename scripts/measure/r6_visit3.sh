#!/bin/bash
# round 6, third visit: the query split of the int8 tier -- tests, A/B on the structured corpora (MDR_MIPS_I8_CB=0 / default), default bench
set -u
TAG=${1:-r06v3}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest (MIPS + CLI)"
timeout 1800 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_gpu.py tests/test_mips_fullsize_gpu.py tests/test_cli_reference_gpu.py tests/test_retrieval_agreement_gpu.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_mips.txt; tail -5 $OUT/pytest_mips.txt
grep -o "query split, nq.*\|case [0-9] beam.*\|clustered 5 M.*" $OUT/pytest_mips.txt | cut -c1-260
echo "== structured A/B"
for CB in 0 1; do
  MDR_MIPS_I8_CB=$CB timeout 900 python bench.py --mode structured > $OUT/structured_cb$CB.json 2> $OUT/structured_cb$CB.err
done
timeout 900 python bench.py --mode structured > $OUT/structured_auto.json 2> $OUT/structured_auto.err
python - $OUT <<'PY'
import json, sys
for tag in ("cb0", "cb1", "auto"):
    try:
        r = json.loads(open(f"{sys.argv[1]}/structured_{tag}.json").read().strip().splitlines()[-1])["structured"]
    except Exception as e:
        print(tag, "failed", e); continue
    for name, v in r.items():
        for nq in ("nq100", "nq200"):
            x = v[nq]
            print(f"{tag:5s} {name:17s} {nq}: {x['ms_per_search']:.4f} ms  int8 decided {x['int8_tier_decided']}  emitted {x['candidates_emitted']:7d} rescored {x['candidates_rescored']:6d}  agree {x['top1_agreement_up_to_exact_ties']}")
PY
echo "== anisotropic + default bench"
SECONDS=0
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench wall seconds: $SECONDS"
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
print("aniso", json.dumps(r.get("anisotropic"))[:1500])
PY
MDR_MIPS_I8_CB=0 timeout 1200 python bench.py --no-cpu-baseline --structured > $OUT/bench_default_cb0.json 2> $OUT/bench_default_cb0.err
python - $OUT/bench_default_cb0.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("CB=0: value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"], r.get("mips_tiers"))
PY
du -sh $OUT

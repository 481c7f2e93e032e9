#!/bin/bash
# Round-3 encoder visit: encoder / gemm parity tests, kernel stats of the default bench. -> gpurun_out/<tag>/
set -u
TAG=${1:-r03e}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest encoder"
timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_gemm_gpu.py tests/test_assemble_gpu.py -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -4
echo "== bench (no cpu baseline)"
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("enc", {k: r["roofline_encoder"][k] for k in ("achieved", "frac")})
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"])
PY
echo "== rocprofv3 kernel stats"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $REPO/bench.py --no-cpu-baseline --no-verify --no-sequential > $OUT/prof.log 2>&1
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/kernel_stats.csv; python - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"]
    for k in ("gemm_big_kernel", "gemm_f16_kernel", "gemm_persist", "attention_stream", "attention_kernel", "attention_cls", "layernorm", "mips_screen8w", "mips_screen8_", "embed_ln"):
        if k in n:
            n = k + n[n.find(k) + len(k):][:28]
            break
    print(f"{n[:60]:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.2f} %")
PY
)
rm -rf $OUT/prof

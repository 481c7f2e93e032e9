#!/bin/bash
# round 6, second visit: the GPU tests that changed, per-kernel times of the k = 1 search on the encoder-geometry corpus, the default bench line
set -u
TAG=${1:-r06v2}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest (changed tests)"
timeout 1800 python -m pytest tests/test_mips_gpu.py tests/test_cli_reference_gpu.py tests/test_cli_gpu.py tests/test_encoder_gpu.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_changed.txt; tail -5 $OUT/pytest_changed.txt
grep -o "case [0-9] beam.*\|fever case.*\|encode_corpus.*vs the reference.*\|--only-eval-ans:.*" $OUT/pytest_changed.txt | cut -c1-260
echo "== rocprofv3 kernel stats: structured encoder-geometry corpus"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_struct -o s -- python $REPO/bench.py --mode structured --structured encoder > $OUT/prof_struct.log 2>&1
S=$(find $OUT/prof_struct -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/structured_encoder_kernel_stats.csv; grep -i "mips\|prep\|final" "$S" | cut -c1-170)
rm -rf $OUT/prof_struct
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_struct -o s -- python $REPO/bench.py --mode structured --structured clustered > $OUT/prof_struct2.log 2>&1
S=$(find $OUT/prof_struct -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/structured_clustered_kernel_stats.csv; grep -i "mips\|prep\|final" "$S" | cut -c1-170)
rm -rf $OUT/prof_struct
cd $REPO
echo "== bench DEFAULT"
SECONDS=0
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench wall seconds: $SECONDS"
python - $OUT/bench_default.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "stage", r["stage_ms"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "traffic_fresh")})
print("seq", r["sequential"]["value"], r["sequential"]["stage_ms"], r["sequential"]["mips_roofline"]["frac"])
print("self_check", r["self_check"]["full_size_exact"], r.get("mips_tiers"))
print("structured", json.dumps(r.get("structured"))[:1500])
PY
du -sh $OUT

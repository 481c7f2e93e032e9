#!/bin/bash
# epilogue rewrite of the 256x256 GEMM kernels: parity, then the bench and in-forward kernel averages
cd $GRAFT_REPO_ROOT; REPO=$(pwd); export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py -q -m gpu 2>&1 | tail -4
for r in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 30 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'hop2_encode', r['stage_ms']['hop2_encode'], 'seq', r['sequential']['value'], r['sequential']['stage_ms']['hop1_encode'], r['sequential']['stage_ms']['hop2_encode'])"; done
cd /tmp; OUT=$REPO/gpurun_out/epi; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $REPO/bench.py --no-cpu-baseline --no-verify --no-sequential --no-anisotropic > $OUT/prof.log 2>&1
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/kernel_stats.csv
python - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.2f} %")
PY

#!/bin/bash
# round 5: mips_refine8_kernel filter-then-rescore: parity tests + per-kernel times of the MIPS-only pipelined and sequential loops (rocprofv3 --kernel-trace --stats)
set -u
TAG=${1:-r5refine}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_gpu.py tests/test_mips_fullsize_gpu.py -x -q > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
cd /tmp
for MODE in "" "--sequential"; do
  rm -rf /tmp/pr
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o b -- python $REPO/bench.py --rows 5000000 --steps 20 --warmup 3 --no-encoder --no-cpu-baseline --no-sequential --no-verify $MODE > $OUT/log$MODE.txt 2>&1
  S=$(find /tmp/pr -name "*kernel_stats.csv" | head -1)
  python - "$S" "loop${MODE}" <<'PY' | tee -a $OUT/kernels.txt
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(mips_\w+<[^>]*>|mips_\w+|prep_queries\w+|finalize\w+)', r['Name'])
    if m and float(r['AverageNs']) > 6000:
        print(f"{sys.argv[2]:18s} {m.group(1):36s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done

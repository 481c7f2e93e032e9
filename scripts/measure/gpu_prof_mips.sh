#!/bin/bash
# per-kernel time of one MIPS-only bench run (rocprofv3 --kernel-trace --stats) -> gpurun_out/<tag>/
set -u
TAG=${1:-prof}; ROWS=${2:-2000000}; BATCH=${3:-100}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $REPO/bench.py --rows $ROWS --steps 20 --warmup 3 --no-encoder --no-cpu-baseline --batch $BATCH > $OUT/stats.log 2>&1
grep '"metric"' $OUT/stats.log | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['roofline'])"
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/kernel_stats_${ROWS}_${BATCH}.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_${ROWS}_${BATCH}.csv")))
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
rm -rf $OUT/stats

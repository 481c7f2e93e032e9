#!/bin/bash
# per-kernel times of one k = 1 search call (planted queries, 5 M rows) with and without the int8 tier
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i8p -o r -- python $GRAFT_REPO_ROOT/scripts/measure/gpu_i8_quick.py 5000000 > /tmp/i8p.log 2>&1
S=$(find /tmp/i8p -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'mdr::' in n or 'fill' in n.lower():
        short = n.replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:60]
        print(f"{int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}  {short}")
PY

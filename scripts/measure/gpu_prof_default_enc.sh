#!/bin/bash
# per-kernel times of the encoder kernels in the default bench
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-verify --no-sequential > /tmp/pe.log 2>&1
S=$(find /tmp/pe -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'mdr' in r['Name'])
for r in rows:
    n=r['Name']
    if 'mdr' in n and 'mips' not in n and 'convert' not in n and 'row_norm' not in n:
        short=n.replace('_ZN3mdr12_GLOBAL__N_1','').replace('(anonymous namespace)::','').replace('void ','')[:46]
        print(f"{short:48s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.2f} ms")
PY

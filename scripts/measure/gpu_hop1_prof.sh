#!/bin/bash
# per-kernel times of an isolated hop-1-shaped forward (100 questions of 8-40 tokens, padded to 70)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ph
ENC_LEN=8,40 ENC_L=70 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -o b -- python $GRAFT_REPO_ROOT/scripts/measure/gpu_enc_forward.py > /tmp/ph.log 2>&1
grep "^tokens" /tmp/ph.log
S=$(find /tmp/ph -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'mdr' in n:
        short=n.replace('_ZN3mdr12_GLOBAL__N_1','').replace('(anonymous namespace)::','').replace('void ','')[:60]
        print(f"{short:62s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.2f} ms")
        tot+=float(r['TotalDurationNs'])
print("sum of kernel time per forward (23 forwards):", tot/23/1e6, "ms")
PY

#!/bin/bash
# Per-kernel attention time of variant builds (e.g. -DMDR_ATTN_ABL=4 --out=libmdrhip_attn4.so) on hop-2-shaped forwards
# (scripts/measure/gpu_enc_forward.py): scripts/measure/gpu_attn_ab.sh libA.so libB.so ...
set -u
REPO=$(pwd)
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pa
  MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o b -- python $REPO/scripts/measure/gpu_enc_forward.py > /tmp/pa.log 2>&1
  grep "^tokens" /tmp/pa.log || tail -5 /tmp/pa.log
  S=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
  python - "$S" "$v" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attention_' in r['Name'] and 'cls' not in r['Name']:
        print(f"{sys.argv[2]:24s} {r['Name'][22:52]} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done

#!/bin/bash
# LayerNorm time per library variant on hop-2-shaped forwards: scripts/measure/gpu_ln_ab.sh libA.so libB.so ...
set -u
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pl
  MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o b -- python $REPO/scripts/measure/gpu_enc_forward.py > /tmp/pl.log 2>&1
  grep "^tokens" /tmp/pl.log || tail -3 /tmp/pl.log
  S=$(find /tmp/pl -name "*kernel_stats.csv" | head -1)
  python - "$S" "$v" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'layernorm_kernel' in r['Name'] or 'gemm_quad' in r['Name']:
        print(f"{sys.argv[2]:24s} {r['Name'][:44]:44s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us  max {float(r['MaxNs'])/1e3:8.2f}")
PY
done

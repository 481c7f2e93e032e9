#!/bin/bash
set -u
TAG=${1:-pmcl2}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for PMC in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/bench.py --rows 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
  P=$(find $OUT/p -name "*counter_collection.csv" | head -1)
  [ -n "$P" ] && python - "$P" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    key = None
    if "gemm_persist" in n: key = "persist " + n[n.index("gemm_persist_kernelILi")+22:][:1]
    elif "attention_kernelILi24" in n: key = "attention24"
    elif "layernorm" in n: key = "layernorm"
    if key: agg[key][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k in sorted(agg):
    for c, v in agg[k].items():
        v.sort(key=lambda x: -x[1]); top = v[: max(1, len(v) // 3)]
        print(f"{k:12s} {c:30s} mean {sum(x[0] for x in top)/len(top):.5g}  dur_us {sum(x[1] for x in top)/len(top)/1e3:.1f} n={len(top)}")
PY
  tail -2 $OUT/log.txt | grep -i -E "error|invalid" | head -2
  rm -rf $OUT/p
done

#!/bin/bash
# Effective shader clock of the persistent GEMM kernels: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel duration.
# usage: scripts/measure/gpu_gemm_clock.sh <tag> <kernel ids...>
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o g -- python $REPO/scripts/measure/gpu_gemm_bench.py 20611 "$@" > $OUT/log.txt 2>&1
F=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
order = []
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_" not in k: continue
    short = k.split("gemm_")[1][:22]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else None
    key = (short, r.get("Dispatch_Id") and 0)
    acc[short].append((float(r["Counter_Value"]), dur, int(r["Dispatch_Id"])))
print(list(rows[0].keys()))
for k, v in acc.items():
    # consecutive launches of the same kernel over different shapes: group by order of appearance in blocks of 23
    v.sort(key=lambda t: t[2])
    for b in range(0, len(v), 23):
        blk = v[b:b + 23][3:]
        cyc = sum(t[0] for t in blk) / len(blk) / 8
        if blk[0][1] is not None:
            dur = sum(t[1] for t in blk) / len(blk)
            print(f"{k:24s} block {b // 23}: cycles/XCD {cyc:9.0f}  duration {dur / 1e3:7.1f} us  clock {cyc / dur:5.2f} GHz")
        else:
            print(f"{k:24s} block {b // 23}: cycles/XCD {cyc:9.0f}")
PY
rm -rf $OUT/p

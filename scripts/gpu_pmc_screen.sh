#!/bin/bash
# HBM traffic (FETCH_SIZE) and SQ counters of the MIPS screen kernel at 5M rows -> gpurun_out/<tag>/
set -u
TAG=${1:-pmcs}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  N=$(echo $PMC | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/bench.py --rows 5000000 --steps 3 --warmup 1 --no-encoder --no-cpu-baseline > $OUT/log_$N.txt 2>&1
  P=$(find $OUT/p -name "*counter_collection.csv" | head -1)
  if [ -n "$P" ]; then head -1 "$P" > $OUT/screen_$N.csv; grep -E "mips_(screen|refine)" "$P" >> $OUT/screen_$N.csv; fi
  rm -rf $OUT/p
done
python - $OUT <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/screen_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = ("main" if "<24, 1" in r["Kernel_Name"] else "sample" if "<24, 0" in r["Kernel_Name"] else "refine")
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:7s} {c:28s} mean {sum(v)/len(v):.6g} n={len(v)}")
PY

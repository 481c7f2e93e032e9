#!/bin/bash
# LDS / issue counters of the persistent GEMM kernels at the hop-2 shapes: scripts/measure/gpu_gemm_pmc.sh <tag> <kernel ids...>
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p_$N -o g -- python $REPO/scripts/measure/gpu_gemm_bench.py 20611 "$@" > $OUT/log_$N.txt 2>&1
  F=$(find $OUT/p_$N -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_" not in k: continue
    short = k.split("gemm_")[1][:24]
    acc[(short, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    print(f"{k:26s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
PY
  rm -rf $OUT/p_$N
done

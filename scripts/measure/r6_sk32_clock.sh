#!/bin/bash
# round 6: effective shader clock (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / kernel duration) of mips_screenk32_kernel as built, compute-only (-DMDR_SK32_ABL=1) and
# DMA-only (-DMDR_SK32_ABL=2): is the missing overlap of HBM stream and LDS / MFMA work a clock (power) effect?
set -u
TAG=${1:-r06sk32clk}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for L in product sk32abl1 sk32abl2; do
  if [ $L = product ]; then unset MDR_LIB_PATH; else export MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/libmdrhip_$L.so; fi
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p_$L -o b -- python $REPO/scripts/measure/r6_groups_ab.py 5000000 f32x2h 256:8 160:8 > $OUT/log_$L.txt 2>&1
  F=$(find $OUT/p_$L -name "*counter_collection.csv" | head -1)
  python - "$F" $L <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mips_screenk32_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
n = len(rows) // 2
for i, name in enumerate(("nq256 (8 waves)", "nq160 (5 waves)")):
    blk = rows[i * n:(i + 1) * n][3:]
    cyc = sum(float(r["Counter_Value"]) for r in blk) / len(blk) / 8
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in blk) / len(blk)
    print(f"{sys.argv[2]:9s} {name}: cycles/XCD {cyc:10.0f}  duration {dur / 1e3:8.1f} us  effective clock {cyc / dur:5.2f} GHz  ({len(blk)} launches)")
PY
  rm -rf $OUT/p_$L
done

#!/bin/bash
# round 6: does the fp16 wide screen-k pass overlap its HBM stream with its LDS / MFMA work? -DMDR_SK32_ABL=1 (no DMA behind the prologue) / =2 (DMA only) builds against the
# product kernel; rocprofv3 kernel averages of mips_screenk32_kernel at 5 M rows, nq 256 and nq 128+32 (5 waves), k 8
set -u
TAG=${1:-r06sk32}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for L in product sk32abl1 sk32abl2; do
  if [ $L = product ]; then unset MDR_LIB_PATH; else export MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/libmdrhip_$L.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$L -o b -- python $REPO/scripts/measure/r6_groups_ab.py 5000000 f32x2h 256:8 160:8 32:8 > $OUT/log_$L.txt 2>&1
  S=$(find $OUT/p_$L -name "*kernel_trace.csv" | head -1)
  python - "$S" $L <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mips_screenk32_kernel" in r["Kernel_Name"]]
# launches come in the order of the shapes: 256:8, 160:8, 32:8 -- 13 launches each (3 warm-up + 10 timed)
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = len(dur) // 3
for i, name in enumerate(("nq256 (8 waves)", "nq160 (5 waves)", "nq32 (1 wave)")):
    d = dur[i * n:(i + 1) * n][3:]
    print(f"{sys.argv[2]:9s} {name}: mips_screenk32_kernel avg {sum(d) / len(d):8.1f} us over {len(d)} launches")
PY
  rm -rf $OUT/p_$L
done

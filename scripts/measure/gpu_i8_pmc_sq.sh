#!/bin/bash
# SQ counters (matrix pipe, VALU, LDS, residency) of the int8 screen kernels in the MIPS-only bench loops at 5 M rows
# (pipelined: mips_screen8w_kernel, 200 queries per pass; sequential: mips_screen8_kernel, 100) -> gpurun_out/<tag>/i8_pmc_sq.txt
set -u
TAG=${1:-i8sq}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
: > $OUT/i8_pmc_sq.txt
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  D=/tmp/pq_$$; rm -rf $D
  timeout 400 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $D -o p -- python $REPO/bench.py --rows 5000000 --steps 3 --warmup 1 --no-encoder --no-cpu-baseline --no-verify > /tmp/pq.log 2>&1 || { echo "pass [$PMC] failed: $(tail -2 /tmp/pq.log | tr '\n' ' ')" >> $OUT/i8_pmc_sq.txt; continue; }
  P=$(find $D -name "*counter_collection.csv" | head -1)
  python - "$P" >> $OUT/i8_pmc_sq.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    kind = None
    for tag in ("mips_screen8w_kernel<12, 1", "mips_screen8_kernel<12, 1", "mips_refine8_kernel", "mips_star8_kernel"):
        if tag in n: kind = tag.split("<")[0] + ("<12,1>" if "<" in tag else "")
    if kind is None: continue
    a = agg.setdefault((kind, r["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in agg.items():
    print(f"{k:28s} {c:30s} launches {n:4d}  mean per launch {v / n:16.1f}")
PY
  rm -rf $D
done
cat $OUT/i8_pmc_sq.txt

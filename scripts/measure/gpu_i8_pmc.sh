#!/bin/bash
# LDS / wait counters of the int8 screen kernel (5 M rows, planted queries)
cd /tmp; export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/i8c
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/i8c -o r -- python $GRAFT_REPO_ROOT/scripts/measure/gpu_i8_quick.py 5000000 > /tmp/i8c.log 2>&1
  F=$(find /tmp/i8c -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "mips_screen8_kernel<12, 1" in k or "mips_screen_kernel<24, 1" in k:
        acc[(k.split("::")[-1][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = [x for x in v if x > 0]
    if v: print(f"{k:36s} {c:24s} n={len(v):3d} max={max(v):.4e} median={sorted(v)[len(v)//2]:.4e}")
PY
done

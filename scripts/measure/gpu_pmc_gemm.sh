#!/bin/bash
# SQ counters for the encoder GEMM / attention kernels of one bench run
set -u
TAG=${1:-pmcg}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  N=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  MDR_GEMM_CFG=${2:-0} timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/bench.py --rows 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log_$N.txt 2>&1
  P=$(find $OUT/p -name "*counter_collection.csv" | head -1)
  [ -n "$P" ] && python - "$P" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    key = None
    if "gemm_f16" in n: key = "gemm " + n[n.index("gemm_f16_kernelILi")+18:][:1] + " " + (n[n.index("GemmCfgILi")+10:][:3] if "GemmCfg" in n else "")
    elif "attention" in n: key = "attention"
    elif "mips_screen_kernel<24, 1>" in n: key = "screen"
    if key: agg[key][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k in sorted(agg):
    for c, v in agg[k].items():
        # only the longest-running quarter (hop-2 launches)
        v.sort(key=lambda x: -x[1]); top = v[: max(1, len(v) // 4)]
        print(f"{k:14s} {c:28s} mean {sum(x[0] for x in top)/len(top):.4g}  dur_us {sum(x[1] for x in top)/len(top)/1e3:.1f} n={len(top)}")
PY
  rm -rf $OUT/p
done

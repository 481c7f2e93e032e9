#!/bin/bash
# SQ counters of the attention kernel on hop-2-shaped forwards (scripts/measure/gpu_enc_forward.py), one rocprofv3 --pmc pass per group
# -> gpurun_out/<tag>/attn_pmc.txt
set -u
TAG=${1:-attnpmc}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
: > $OUT/attn_pmc.txt
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  D=/tmp/pp_$$; rm -rf $D
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $D -o p -- python $REPO/scripts/measure/gpu_enc_forward.py > /tmp/pp.log 2>&1 || { echo "pass [$PMC] failed: $(tail -2 /tmp/pp.log | tr '\n' ' ')" >> $OUT/attn_pmc.txt; continue; }
  P=$(find $D -name "*counter_collection.csv" | head -1)
  python - "$P" >> $OUT/attn_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    kind = "attention_stream" if "attention_stream" in n else ("gemm_big<%s>" % n.split("ILi")[1][0] if "gemm_big" in n else ("layernorm" if "layernorm" in n else None))
    if kind is None: continue
    a = agg.setdefault((kind, r["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in agg.items():
    print(f"{k:18s} {c:30s} launches {n:4d}  mean per launch {v / n:14.1f}")
PY
  rm -rf $D
done
cat $OUT/attn_pmc.txt

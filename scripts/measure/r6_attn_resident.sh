#!/bin/bash
# round 6: attention with ring-RESIDENT K / V for sequences of at most 192 keys (-DMDR_ATTN_RESIDENT=1 build) against the product kernel: encoder tests under the
# variant, then alternating rocprofv3 kernel averages inside the default bench command (same box)
set -u
TAG=${1:-r06attn}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$REPO/multihop_dense_retrieval_amd/libmdrhip_attn_resident.so
echo "== encoder tests under the variant"
MDR_LIB_PATH=$V timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -q 2>&1 | tail -3
cd /tmp
for rep in 1 2; do
  for L in product variant; do
    if [ $L = variant ]; then export MDR_LIB_PATH=$V; else unset MDR_LIB_PATH; fi
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$L$rep -o b -- python $REPO/bench.py --no-cpu-baseline --no-sequential > $OUT/bench_$L$rep.json 2> $OUT/bench_$L$rep.err
    S=$(find $OUT/prof_$L$rep -name "*kernel_stats.csv" | head -1)
    python - "$S" $L $rep $OUT/bench_$L$rep.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
att = [r for r in rows if "attention_ring_kernel" in r["Name"]]
r = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:8s} rep {sys.argv[3]}: attention_ring_kernel calls {att[0]['Calls']} avg {float(att[0]['AverageNs']) / 1e3:.2f} us; headline {r['value']} q/s, hop2_encode {r['stage_ms']['hop2_encode']} ms")
PY
    rm -rf $OUT/prof_$L$rep
  done
done

#!/bin/bash
# round 6: the 16-queries-per-wave int8 kernel with hand-placed LDS reads and counted waits in its four MFMA chains (-DMDR_I8N_ASM=1) against the product (hipcc's schedule:
# a full lgkmcnt(0) in front of nearly every MFMA). Sequential loop (100 queries per search), rocprofv3 kernel averages + un-profiled runs, alternating on one box.
# NOTE: the -DMDR_I8N_ASM source was removed again after this measurement (neutral); the chain is archived with re-run notes as scripts/ubench/mdr_mips_chain8x16.inl.txt.
set -u
TAG=${1:-r06i8n}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$REPO/multihop_dense_retrieval_amd/libmdrhip_i8nasm.so
echo "== correctness under the variant"
MDR_LIB_PATH=$V timeout 1200 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_gpu.py -m gpu -q -x 2>&1 | tail -3
cd /tmp
for rep in 1 2; do
  for L in product variant; do
    if [ $L = variant ]; then export MDR_LIB_PATH=$V; else unset MDR_LIB_PATH; fi
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$L$rep -o b -- python $REPO/bench.py --no-encoder --sequential --no-cpu-baseline --steps 40 > $OUT/bench_$L$rep.json 2> $OUT/bench_$L$rep.err
    S=$(find $OUT/p_$L$rep -name "*kernel_stats.csv" | head -1)
    python - "$S" $L $rep $OUT/bench_$L$rep.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = [r for r in rows if "mips_screen8_kernel<12, 1" in r["Name"]]
r = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:8s} rep {sys.argv[3]}: mips_screen8 main pass calls {w[0]['Calls']} avg {float(w[0]['AverageNs']) / 1e3:.1f} us; MIPS-only sequential loop {r['value']} q/s, search stages {r['stage_ms']['hop1_search']} + {r['stage_ms']['hop2_search']} ms, roofline frac {r['roofline']['frac']}, exact {r['self_check'].get('full_size_exact')}")
PY
    rm -rf $OUT/p_$L$rep
  done
done

#!/bin/bash
# round 6: the wide int8 kernel with every `bounds` slice pinned behind its MFMA (-DMDR_I8W_PIN=1: hipcc had sunk the eight slices into one block between MFMA 8 and 9) against the product
set -u
TAG=${1:-r06pin}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V=$REPO/multihop_dense_retrieval_amd/libmdrhip_i8wpin.so
echo "== correctness under the variant"
MDR_LIB_PATH=$V timeout 1200 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_gpu.py -m gpu -q -x 2>&1 | tail -3
cd /tmp
for rep in 1 2; do
  for L in product variant; do
    if [ $L = variant ]; then export MDR_LIB_PATH=$V; else unset MDR_LIB_PATH; fi
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$L$rep -o b -- python $REPO/bench.py --no-cpu-baseline --no-sequential > $OUT/bench_$L$rep.json 2> $OUT/bench_$L$rep.err
    S=$(find $OUT/p_$L$rep -name "*kernel_stats.csv" | head -1)
    python - "$S" $L $rep $OUT/bench_$L$rep.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = [r for r in rows if "mips_screen8w_kernel<12, 1" in r["Name"]]
r = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:8s} rep {sys.argv[3]}: mips_screen8w main pass calls {w[0]['Calls']} avg {float(w[0]['AverageNs']) / 1e3:.1f} us; headline {r['value']} q/s, hop2_search {r['stage_ms']['hop2_search']} ms, exact {r['self_check']['full_size_exact']}")
PY
    rm -rf $OUT/p_$L$rep
  done
done
unset MDR_LIB_PATH
echo "== un-profiled headline, alternating"
for rep in 1 2; do
  for L in product variant; do
    if [ $L = variant ]; then export MDR_LIB_PATH=$V; else unset MDR_LIB_PATH; fi
    timeout 600 python $REPO/bench.py --no-cpu-baseline --no-sequential > $OUT/plain_$L$rep.json 2> /dev/null
    python -c "
import json
r=json.loads(open('$OUT/plain_$L$rep.json').read().strip().splitlines()[-1])
print('$L rep $rep: headline', r['value'], 'q/s, ms/step', r['ms_per_step'], 'search', r['stage_ms']['hop2_search'], 'roofline frac', r['roofline']['frac'])"
  done
done

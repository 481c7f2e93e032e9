#!/bin/bash
# round 6 experiment: the int8 main pass for > 128 queries on FOUR waves of 64 queries (mdr_mips_screen_i8q.inl, -DMDR_I8Q=1 build, MDR_MIPS_I8Q=1) against the eight-wave kernel
# NOTE: the kernel lives in scripts/ubench/mdr_mips_screen_i8q.inl.txt now (archived after this measurement); to re-run, restore it into csrc/ with the hook in run_screen8w (git show 3016e19^) and build -DMDR_I8Q=1.
set -u
TAG=${1:-r06i8q}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/libmdrhip_i8q.so
echo "== correctness under the variant (MDR_MIPS_I8Q=1)"
MDR_MIPS_I8Q=1 timeout 1200 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_gpu.py -m gpu -q -x 2>&1 | tail -4
echo "== timing, same library, alternating"
for rep in 1 2; do
  for Q in 0 1; do
    MDR_MIPS_I8Q=$Q timeout 300 python scripts/measure/r6_groups_ab.py 5000000 f32x2h 200:1 256:1 800:1 2>&1 | grep groups= | sed "s/^/I8Q=$Q /" | tee -a $OUT/timing.txt
  done
done
echo "== MIPS-only bench (pipelined loop: fused 200-query pass), kernel averages"
cd /tmp
for Q in 0 1; do
  MDR_MIPS_I8Q=$Q timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$Q -o b -- python $REPO/bench.py --no-encoder --no-cpu-baseline --no-sequential --steps 40 > $OUT/bench_q$Q.json 2> $OUT/bench_q$Q.err
  S=$(find $OUT/prof$Q -name "*kernel_stats.csv" | head -1)
  python - "$S" $Q $OUT/bench_q$Q.json <<'PY'
import csv, json, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mips_screen8" in r["Name"]:
        print(f"I8Q={sys.argv[2]} {r['Name'].split('(')[0][-48:]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
r = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"I8Q={sys.argv[2]} bench value {r['value']} ms/step {r['ms_per_step']} exact {r['self_check'].get('full_size_exact')} stage {r['stage_ms']}")
PY
  rm -rf $OUT/prof$Q
done

#!/bin/bash
# round 5: `scripts/measure/bench_loops.py --loop shift` (the hop-1 forward of batch i+2 beside the corpus pass of step i; the hop-2 forward alone) against the default pipelined loop
# (hop 1 beside hop 2), with the product library and with a -DMDR_I8W_SLOTS=2 build of the wide int8 screen (97 instead of 145 KiB of LDS per CU, so that the
# hop-1 forward's 64x64 GEMM blocks can share a CU with it). Alternating runs on ONE box -> gpurun_out/<tag>/shift.txt
set -u
TAG=${1:-r5shift}; OUT=gpurun_out/$TAG; mkdir -p $OUT; REPO=$(pwd)
for rep in 1 2; do
  for V in "pipelined libmdrhip.so" "shift libmdrhip.so" "pipelined libmdrhip_i8w2.so" "shift libmdrhip_i8w2.so"; do
    set -- $V; LOOP=$1; LIB=$2
    MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$LIB timeout 300 python scripts/measure/bench_loops.py --loop $LOOP --no-cpu-baseline --no-anisotropic --no-sequential > $OUT/b_${LOOP}_${LIB}_$rep.json 2> $OUT/b_${LOOP}_${LIB}_$rep.err || tail -3 $OUT/b_${LOOP}_${LIB}_$rep.err
    python - $OUT/b_${LOOP}_${LIB}_$rep.json "$LOOP $LIB" <<'PY' | tee -a $OUT/shift.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:32s}: {r['value']:9.1f} q/s  {r['ms_per_step']:.3f} ms  stage {r['stage_ms']}  exact {r['self_check']['full_size_exact']}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
  done
done

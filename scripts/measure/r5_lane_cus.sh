#!/bin/bash
# round 5: CU-partitioned encoder lanes in the headline loop -- the side lane (hop-1 forward of the next batch) on its own CUs, the hop-2 forward on the rest
# (mdr_stream_create_cu_range; scripts/measure/bench_loops.py --lane-cus N with a -DMDR_CU_LANES=1 build via MDR_LIB_PATH). Alternating runs on ONE box -> gpurun_out/<tag>/lanes.txt
set -u
TAG=${1:-r5lanes}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for N in 0 16 32 48 64; do
    timeout 300 python scripts/measure/bench_loops.py --lane-cus $N --no-cpu-baseline --no-anisotropic --no-sequential > $OUT/b_${N}_$rep.json 2> $OUT/b_${N}_$rep.err || tail -3 $OUT/b_${N}_$rep.err
    python - $OUT/b_${N}_$rep.json $N <<'PY' | tee -a $OUT/lanes.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"lane_cus {int(sys.argv[2]):3d}: {r['value']:9.1f} q/s  {r['ms_per_step']:.3f} ms  stage {r['stage_ms']}  exact {r['self_check']['full_size_exact']}")
except Exception as e:
    print("lane_cus", sys.argv[2], "failed:", e)
PY
  done
done

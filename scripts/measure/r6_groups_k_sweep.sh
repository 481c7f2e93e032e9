#!/bin/bash
# round 6: old cut (MDR_MIPS_EVEN_GROUPS=0) vs even cut (=1) of the query groups over k, at BASELINE configs[4]'s shard shape and at 5 M fp32-accurate rows
set -u
TAG=${1:-r06gk}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
for M in 0 1; do
  MDR_MIPS_EVEN_GROUPS=$M timeout 600 python scripts/measure/r6_groups_ab.py 6250000 bf16 800:8 800:16 800:32 800:64 800:100 800:250 2>&1 | grep groups= | tee -a $OUT/sweep.txt
  MDR_MIPS_EVEN_GROUPS=$M timeout 600 python scripts/measure/r6_groups_ab.py 5000000 f32x2h 800:8 800:32 800:100 500:5 300:100 2>&1 | grep groups= | tee -a $OUT/sweep.txt
done
done

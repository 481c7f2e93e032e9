#!/bin/bash
# 128- vs 256-queries-per-pass MIPS kernels at 5M rows: timing sweep over nq (k = 1 and k = 8)
for nq in 100 128 200 256 400 800; do
  for w in 0 1; do
    for k in ${KS:-1}; do
      echo -n "wide=$w "; MDR_MIPS_WIDE=$w SWEEP_NQ=$nq SWEEP_K=$k SWEEP_PLANTED=${PLANTED:-0} timeout 300 python scripts/measure/gpu_ksweep.py 5000000 ${STORAGE:-f32x2h} 2>&1 | grep screen
    done
  done
done

#!/bin/bash
# round 5: wide int8 screen with two LDS slots (-DMDR_I8W_SLOTS=2 -> libmdrhip_i8w2.so) against three, MIPS-only loop, alternating runs: no difference (1.165-1.176 ms both)
REPO=$(pwd)
for rep in 1 2 3; do for LIB in libmdrhip.so libmdrhip_i8w2.so; do
MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$LIB python bench.py --rows 5000000 --steps 40 --warmup 5 --no-encoder --no-cpu-baseline --no-verify --no-sequential 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LIB', r['ms_per_step'], r['roofline']['avg_launch_ms'])"
done; done

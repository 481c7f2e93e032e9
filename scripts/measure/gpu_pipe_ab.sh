#!/bin/bash
# A/B of variant builds inside the real retrieval loop (1 M rows: the encoder does not care): scripts/measure/gpu_pipe_ab.sh libA.so libB.so ...
set -u
REPO=$(pwd)
for r in 1 2; do
  for v in "$@"; do
    echo "-- $v round $r"
    MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$v timeout 250 python bench.py --rows 1000000 --no-cpu-baseline --no-verify --no-sequential 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
  done
done

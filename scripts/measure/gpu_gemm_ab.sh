#!/bin/bash
# A/B of variant builds of the encoder GEMMs on one box, interleaved: scripts/measure/gpu_gemm_ab.sh <M> <kernel list> -- libA.so libB.so ...
set -u
M=$1; shift
KS=""
while [ "$1" != "--" ]; do KS="$KS $1"; shift; done
shift
REPO=$(pwd)
for r in 1 2; do
  for v in "$@"; do
    echo "-- $v round $r"
    MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$v timeout 200 python scripts/measure/gpu_gemm_bench.py $M $KS 2>&1 | grep -v amdgpu.ids
  done
done

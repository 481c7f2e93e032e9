#!/bin/bash
# One GPU-box visit of A/B measurements (no parity tests): everything lands in gpurun_out/<tag>/.
# usage: scripts/measure/gpu_ab.sh <tag>
#   - MIPS corpus-stream cache policy: default vs nt build of the same sources (MDR_LIB_PATH selects the variant library)
#   - encoder GEMM kernels at the hop-1 (2.4 k rows) and hop-2 (20.6 k rows) shapes
set -u
TAG=${1:-ab}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.txt 2>&1

echo "== MIPS 5M nq=100 k=1: default vs nt (interleaved, 3 rounds)"
for r in 1 2 3; do
  for v in libmdrhip.so libmdrhip_nt.so; do
    [ -f $REPO/multihop_dense_retrieval_amd/$v ] || continue
    echo "-- $v round $r"
    MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/$v SWEEP_NQ=100 SWEEP_K=1 SWEEP_PLANTED=1 timeout 300 python scripts/measure/gpu_ksweep.py 5000000 2>&1 | grep screen
  done
done | tee $OUT/mips_nt_ab.txt

echo "== GEMM kernels, hop-1 shape (M=2400): 1=64x64 2=128x128 4=persistent 256x128 6=persistent 256x256"
timeout 300 python scripts/measure/gpu_gemm_bench.py 2400 1 2 4 6 2>&1 | tee $OUT/gemm_m2400.txt
echo "== GEMM kernels, hop-2 shape (M=20611)"
timeout 300 python scripts/measure/gpu_gemm_bench.py 20611 4 6 2>&1 | tee $OUT/gemm_m20611.txt
du -sh $OUT

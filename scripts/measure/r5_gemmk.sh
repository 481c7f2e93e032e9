#!/bin/bash
# round 5: the GEMM-structured screen-k main pass (mips_gemmk_kernel) against the kernel it replaces (MDR_MIPS_GEMMK=0): parity tests, then timings at the
# shapes VERDICT r4 item 5 names -- 6.25 M bf16 rows, nq 800, k 8 / 100 / 250; 5 M fp32-accurate rows, nq 300 k 8, nq 256 k 8 -> gpurun_out/<tag>/
set -u
# round 6: the kernel lives in -DMDR_MIPS_GEMMK=1 builds only:  python -m multihop_dense_retrieval_amd.build -DMDR_MIPS_GEMMK=1 --out=libmdrhip_gemmk.so
export MDR_LIB_PATH=${MDR_LIB_PATH:-$PWD/multihop_dense_retrieval_amd/libmdrhip_gemmk.so}
TAG=${1:-r5gemmk}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_mips_gpu.py -x -q -k "screenk or bf16_nq800 or adversarial or topk" > $OUT/pytest.txt 2>&1; echo rc=$? >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
for G in 1 0; do
  for K in 8 100 250; do
    MDR_MIPS_GEMMK=$G SWEEP_NQ=800 SWEEP_K=$K timeout 300 python scripts/measure/gpu_ksweep.py 6250000 bf16 2>&1 | grep screen | sed "s/^/GEMMK=$G /" | tee -a $OUT/sweep.txt
  done
  for NQ in 256 300; do
    MDR_MIPS_GEMMK=$G SWEEP_NQ=$NQ SWEEP_K=8 timeout 300 python scripts/measure/gpu_ksweep.py 5000000 2>&1 | grep screen | sed "s/^/GEMMK=$G /" | tee -a $OUT/sweep.txt
  done
done

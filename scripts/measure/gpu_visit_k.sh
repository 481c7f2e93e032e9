#!/bin/bash
# GPU visit: MIPS parity tests (incl. the k > 1 screen), k sweep, 2-rank weak/strong bench on one shared GPU (gloo)
set -u
OUT=gpurun_out/${1:-visit_k}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mips_gpu.py -x -q 2>&1 | tail -25 > $OUT/pytest_mips.txt; tail -12 $OUT/pytest_mips.txt
timeout 300 python scripts/measure/gpu_ksweep.py 5000000 > $OUT/ksweep_5m.txt 2>&1; cat $OUT/ksweep_5m.txt | tail -20
timeout 300 python scripts/measure/gpu_ksweep.py 1000000 bf16 > $OUT/ksweep_1m_bf16.txt 2>&1; tail -20 $OUT/ksweep_1m_bf16.txt
for sc in weak strong; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-gpu --scaling $sc --rows 2000000 > $OUT/two_rank_$sc.log 2>&1
  tail -1 $OUT/two_rank_$sc.log | cut -c1-1500
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-gpu --scaling weak --rows 2000000 --no-encoder > $OUT/two_rank_weak_mips.log 2>&1
tail -1 $OUT/two_rank_weak_mips.log | cut -c1-1500

#!/bin/bash
# Two ranks sharing ONE GPU over gloo (--share-gpu): executes every N > 1 code path of bench.py (row sharding, embedding
# all-gather, fused per-shard search, list exchange + merge, weak + strong sub-result, pipelined + sequential loops) where only a
# 1-GPU box is available. Not a performance number. Also a one-rank RCCL self-test.
TAG=${1:-two_rank}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for extra in "" "--beam 2 --topk 2" "--no-encoder" "--scaling strong" "--sequential"; do
  echo "== bench --gpus 2 (gloo, shared GPU) $extra"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 \
      --backend gloo --share-gpu --rows 1000000 --steps 3 --warmup 2 $extra 2> $OUT/err.txt | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print({k: r.get(k) for k in ('value', 'ms_per_step', 'n_gpus', 'scaling')}, r['self_check'].get('full_size_exact'), r['config']['global_batch'], 'strong' in str(r.get('strong_scaling')), r.get('sequential', {}).get('value'), r.get('strong_scaling', {}).get('value'))
" || tail -5 $OUT/err.txt
done
echo "== same, one rank (reference ids)"
python bench.py --rows 1000000 --steps 3 --warmup 2 --no-encoder --no-cpu-baseline --sequential --dump-ids $OUT/one.npz > /dev/null 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --backend gloo --share-gpu \
    --rows 1000000 --steps 3 --warmup 2 --no-encoder --scaling strong --sequential --dump-ids $OUT/two.npz > /dev/null 2>&1
python -c "
import numpy as np
a, b = np.load('$OUT/one.npz'), np.load('$OUT/two.npz')
print('two-shard ids == one-index ids:', all(np.array_equal(a[k], b[k]) for k in ('I', 'I2')), 'max score diff', max(float(np.abs(a[k] - b[k]).max()) for k in ('D', 'D2')))
"

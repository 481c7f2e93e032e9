#!/bin/bash
# BASELINE.json configs[3] / configs[4] as ONE shard sees them (the 8-GPU runs themselves need an 8-GPU node):
#   configs[3]: 5.23M-row fp32 index row-sharded over 8 GPUs -> 654k rows per shard, beam 4, topk 4, 8 x 100 questions' queries per shard
#   configs[4]: 50M-row bf16 index over 8 GPUs -> 6.25M rows per shard, beam 8, k = 100 as the MIPS stress size (chain topk clamped to beam^2 = 64)
OUT=gpurun_out/${1:-configs}; mkdir -p $OUT
echo "== configs[3] shard: 654167 rows f32, beam 4 topk 4, batch 100"
timeout 900 python bench.py --rows 654167 --beam 4 --topk 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/config3_shard.json 2> $OUT/config3.err; tail -2 $OUT/config3.err
echo "== configs[4] shard: 6.25M rows bf16, beam 8 topk 64, batch 100"
timeout 900 python bench.py --rows 6250000 --storage bf16 --beam 8 --topk 64 --steps 5 --warmup 3 --no-cpu-baseline > $OUT/config4_shard.json 2> $OUT/config4.err; tail -2 $OUT/config4.err
echo "== configs[4] MIPS stress: 6.25M bf16, nq 800, k 100"
SWEEP_NQ=800 SWEEP_K=100 timeout 600 python scripts/measure/gpu_ksweep.py 6250000 bf16 2>&1 | grep screen | tee $OUT/config4_mips_k100.txt
python - $OUT <<'PY'
import json, sys
for n in ("config3_shard", "config4_shard"):
    try:
        r = json.loads(open(f"{sys.argv[1]}/{n}.json").readline())
        print(n, r["value"], "q/s", r["ms_per_step"], "ms", r["stage_ms"], "exact:", r["self_check"]["full_size_exact"], "seq:", r.get("sequential", {}).get("value"), r["roofline"]["kernel"], r["roofline"]["corpus_passes_per_launch"])
    except Exception as e:
        print(n, "failed", e)
PY

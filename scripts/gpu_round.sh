#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprofv3 summaries. Everything lands in gpurun_out/.
# usage: scripts/gpu_round.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
EXTRA="$@"
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print('torch', torch.__version__, 'gpu', torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.txt 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/env.txt

echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest_gpu.txt; tail -15 $OUT/pytest_gpu.txt

echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -5 $OUT/smoke.txt

echo "== bench 1M (MIPS only)"
timeout 600 python bench.py --rows 1000000 --steps 50 --warmup 5 --no-encoder --no-cpu-baseline $EXTRA > $OUT/bench_1m_mips.json 2> $OUT/bench_1m_mips.err; cat $OUT/bench_1m_mips.json; tail -3 $OUT/bench_1m_mips.err
echo "== bench 5M (MIPS only)"
timeout 900 python bench.py --rows 5000000 --steps 20 --warmup 3 --no-encoder --no-cpu-baseline $EXTRA > $OUT/bench_5m_mips.json 2> $OUT/bench_5m_mips.err; cat $OUT/bench_5m_mips.json; tail -3 $OUT/bench_5m_mips.err
echo "== bench DEFAULT (5M, 2-hop with encoder, cpu baseline) -- the headline line"
timeout 900 python bench.py $EXTRA > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json; tail -3 $OUT/bench_default.err

echo "== rocprofv3 kernel stats (default bench command)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $REPO/bench.py --no-cpu-baseline $EXTRA > $OUT/prof_stats.log 2>&1
S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/kernel_stats.csv; cut -c1-150 "$S" | head -14)
rm -rf $OUT/prof_stats
echo "== rocprofv3 pmc FETCH_SIZE (1M, MIPS only)"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_pmc -o mips1m -- python $REPO/bench.py --rows 1000000 --steps 3 --warmup 1 --no-encoder --no-cpu-baseline $EXTRA > $OUT/prof_pmc.log 2>&1
P=$(find $OUT/prof_pmc -name "*counter_collection.csv" | head -1); [ -n "$P" ] && (head -1 "$P" > $OUT/pmc_fetch_size.csv; grep -E "mips_(screen|screen8|screen8w|screen32|stream)_kernel" "$P" >> $OUT/pmc_fetch_size.csv; head -3 $OUT/pmc_fetch_size.csv | cut -c1-300)
rm -rf $OUT/prof_pmc
# keep the transfer small: drop the big traces, keep csv summaries
find $OUT -name "*.db" -size +20M -delete 2>/dev/null
du -sh $OUT

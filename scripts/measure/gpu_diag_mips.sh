#!/bin/bash
# MIPS kernel diagnostics: query-count sweep (is the stream kernel compute- or HBM-side bound?) + SQ counters.
set -u
TAG=${1:-diag}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for B in 16 48 64 100 128; do
  timeout 300 python bench.py --rows 2000000 --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --batch $B 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('batch',$B, 'search_ms', r['roofline']['avg_launch_ms'], 'GB/s', r['roofline']['achieved'])" | tee -a $OUT/sweep.txt
done
cd /tmp
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_$N -o p -- python $REPO/bench.py --rows 2000000 --steps 3 --warmup 1 --no-encoder --no-cpu-baseline > $OUT/pmc_$N.log 2>&1
  P=$(find $OUT/pmc_$N -name "*counter_collection.csv" | head -1)
  if [ -n "$P" ]; then head -1 "$P" > $OUT/pmc_$N.csv; grep mips_stream "$P" >> $OUT/pmc_$N.csv; fi
  rm -rf $OUT/pmc_$N
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $REPO/bench.py --rows 2000000 --steps 10 --warmup 2 --no-encoder --no-cpu-baseline > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -exec sh -c 'head -1 {} > '$OUT'/kernel_trace_head.csv; grep mips_stream {} | head -5 >> '$OUT'/kernel_trace_head.csv' \;
rm -rf $OUT/stats
ls -la $OUT; cat $OUT/sweep.txt; head -5 $OUT/kernel_stats.csv

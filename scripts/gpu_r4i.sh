#!/bin/bash
set -u
TAG=${1:-r04i}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-anisotropic --no-sequential > $OUT/prof_stats.log 2>&1
S=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && (cp "$S" $OUT/kernel_stats.csv; cut -c1-170 "$S" | head -24)
rm -rf $OUT/prof_stats

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the large-M GEMM kernels in isolation (scripts/measure/gpu_gemm_bench.py) -> gpurun_out/<tag>/
set -u
TAG=${1:-pmcg}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for PMC in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p_$PMC -o p -- python $REPO/scripts/measure/gpu_gemm_bench.py 20611 4 6 > $OUT/$PMC.log 2>&1
  P=$(find $OUT/p_$PMC -name "*counter_collection.csv" | head -1)
  python - "$P" $PMC <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm" not in n: continue
    key = (("big" if "gemm_big" in n else "persist") + " EPI" + n.split("ILi")[1][0], r["Grid_Size"])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (c, v) in agg.items():
    print(f"{sys.argv[2]:10s} {k[0]:14s} launches {c:3d}  mean {v / c / 1024:9.1f} MiB (counter KB / 1024; FETCH_SIZE reads 1/2 of streamed bytes on gfx950)")
PY
  rm -rf $OUT/p_$PMC
done

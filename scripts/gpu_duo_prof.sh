#!/bin/bash
# per-kernel times inside the real forward, default heuristic vs MDR_GEMM_CFG=60
cd $GRAFT_REPO_ROOT; REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for cfg in 0 60; do
  OUT=$REPO/gpurun_out/duoprof_$cfg; rm -rf $OUT; mkdir -p $OUT
  MDR_GEMM_CFG=$cfg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $REPO/bench.py --no-cpu-baseline --no-verify --no-sequential --no-anisotropic > $OUT/prof.log 2>&1
  S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/kernel_stats.csv
  echo "== MDR_GEMM_CFG=$cfg"; python - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"]
    print(f"{n[:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.2f} %")
PY
done

#!/bin/bash
# per-kernel time of the default bench (rocprofv3 --kernel-trace --stats) -> gpurun_out/<tag>/kernel_stats.csv + top list
set -u
TAG=${1:-profb}; shift || true; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
grep '"metric"' $OUT/stats.log | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['stage_ms'])"
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/kernel_stats.csv
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1); cp "$T" $OUT/kernel_trace.csv 2>/dev/null
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
python - "$OUT/kernel_trace.csv" <<'PY'
# hop-2 encoder forward of the LAST step: kernels between the last two mips_screen<..,1> launches
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mips_screen_kernel<24, 1>" in r["Kernel_Name"]]
if len(idx) >= 2:
    seg = rows[idx[-2] + 1: idx[-1]]
    agg = collections.OrderedDict()
    for r in seg:
        n = r["Kernel_Name"][:70]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 if seg else 0
    print(f"-- between the two searches of the last step: {len(seg)} kernels, span {span:.0f} us, busy {sum(a[1] for a in agg.values()):.0f} us")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {n:70s} x{c:4d} total {t:8.1f} us  avg {t/c:7.1f}")
PY
rm -rf $OUT/stats; gzip -f $OUT/kernel_trace.csv 2>/dev/null

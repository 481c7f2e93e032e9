cd /tmp; export TMPDIR=/tmp
GEMM_TORCH_REF=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/refn -o r -- python $GRAFT_REPO_ROOT/scripts/measure/gpu_gemm_bench.py 20611 6 > /tmp/refn.log 2>&1
S=$(find /tmp/refn -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith('Cijk') or 'Cijk' in r['Name']:
        print(r['Calls'], f"{float(r['AverageNs'])/1e3:.1f}us", r['Name'])
PY

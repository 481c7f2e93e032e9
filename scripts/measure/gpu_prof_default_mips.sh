cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-verify --no-sequential > /tmp/pd.log 2>&1
S=$(find /tmp/pd -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'mips' in n or 'prep_q' in n or 'finalize' in n:
        short=n.replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:48]
        print(f"{short:50s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
grep "^{" /tmp/pd.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"stage_ms\"], d[\"mips_tiers\"])"

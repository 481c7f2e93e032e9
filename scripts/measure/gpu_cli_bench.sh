#!/bin/bash
# round 4: CLI bench only (host pipeline iteration)
set -u
TAG=${1:-r04b}; shift || true
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python bench.py --mode cli "$@" > $OUT/bench_cli.json 2> $OUT/bench_cli.err; grep -v "^\[assets\]\|Avg\|Questions num\|Loading\|Building\|Corpus size\|Encoding\|Evaluating" $OUT/bench_cli.err | tail -15 | cut -c1-300
python - $OUT/bench_cli.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("device_loop", r.get("device_loop"))
for k in ("cli_default", "cli_device"):
    if k in r:
        print(k, {x: r[k][x] for x in ("value", "ms_per_batch", "steady_state_queries_per_s", "whole_process_seconds")}, r[k]["stats"])
print("ratio", r.get("cli_over_device_loop"), "identical", r.get("legs_jsonl_identical"))
PY

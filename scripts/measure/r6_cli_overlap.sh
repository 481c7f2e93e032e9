#!/bin/bash
# round 6: CLI start-up with the index upload on its own thread beside weights + graph captures (MDR_CLI_OVERLAP_UPLOAD=1, default) against the serial order (=0)
set -u
TAG=${1:-r06cli}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== CLI tests"
timeout 1800 python -m pytest tests/test_cli_gpu.py tests/test_cli_reference_gpu.py tests/test_cli_multirank_gpu.py tests/test_bench_selflaunch_gpu.py -m gpu -q 2>&1 | tail -4
echo "== bench --mode cli, overlap on / off (assets kept between the runs)"
for O in 1 0 1 0; do
  MDR_CLI_OVERLAP_UPLOAD=$O timeout 1200 python bench.py --mode cli --cli-keep --cli-dir /dev/shm/mdr_cli_assets --no-sequential > $OUT/cli_o$O.json 2> $OUT/cli_o$O.err
  python - $OUT/cli_o$O.json $O <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("cli_default", "cli_device"):
    if k in r:
        s = r[k]["startup_s"]
        print(f"overlap={sys.argv[2]} {k}: whole process {r[k]['whole_process_seconds']} s, loop {r[k]['loop_seconds']} s, {r[k]['value']} q/s; weights {s.get('device_init_and_weights')} captures {s.get('graph_captures')} index_upload {s.get('index_upload')} arena {s.get('token_arena')}")
print("identical", r.get("legs_jsonl_identical"))
PY
done
rm -rf /dev/shm/mdr_cli_assets
echo "== two ranks on one GPU over gloo (bench.py starts them)"
timeout 900 python bench.py --mode cli --gpus 2 --share-gpu --backend gloo --rows 400000 --questions 1000 --no-sequential --cli-legs default,device > $OUT/cli_2ranks.json 2> $OUT/cli_2ranks.err; tail -2 $OUT/cli_2ranks.err | cut -c1-200
python - $OUT/cli_2ranks.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("n_gpus", r["n_gpus"], {k: {x: r[k][x] for x in ("value", "records", "whole_process_seconds")} for k in ("cli_default", "cli_device") if k in r}, "identical", r.get("legs_jsonl_identical"))
except Exception as e:
    print("2-rank cli legs failed:", e)
PY

set -u
OUT=gpurun_out/cls; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_retrieval_agreement_gpu.py tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -3
REPO=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/p -o p -- python $REPO/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sequential --no-verify > $REPO/$OUT/bench.json 2> $REPO/$OUT/bench.err
S=$(find $REPO/$OUT/p -name "*kernel_stats.csv" | head -1); head -14 $S | cut -c1-200
grep -E "attention_cls|embed_ln" $S | cut -c1-200
tail -1 $REPO/$OUT/bench.json | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['stage_ms'])"
rm -rf $REPO/$OUT/p

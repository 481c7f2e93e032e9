#!/bin/bash
# round 5: the drop-in CLI on 5 M-row assets built ONCE (--cli-keep): the reference's flags at beam 1 x topk 1 (the headline shape), at the reference's own
# argparse defaults (--beam-size 5 --topk 2, eval_mhop_retrieval.py:55,61) and at its downstream setting b50_k50 (README.md:240-241) -> gpurun_out/<tag>/
set -u
TAG=${1:-r5cli}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python bench.py --mode cli --cli-keep > $OUT/cli_b1.json 2> $OUT/cli_b1.err
timeout 600 python bench.py --mode cli --cli-keep --beam 5 --topk 2 --no-sequential > $OUT/cli_b5k2.json 2> $OUT/cli_b5k2.err
timeout 900 python bench.py --mode cli --cli-keep --beam 50 --topk 50 --no-sequential > $OUT/cli_b50k50.json 2> $OUT/cli_b50k50.err
rm -rf /dev/shm/mdr_cli_bench /tmp/mdr_cli_bench

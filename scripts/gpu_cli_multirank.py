#!/usr/bin/env python3
"""Launcher of the multi-rank CLI self-test (tests/test_cli_multirank_gpu.py): runs the drop-in eval_mhop_retrieval.main with the given
arguments under torch.distributed.run and leaves every rank's counters (eval_mhop_retrieval.LAST_RUN: loop time, encoder forwards,
graph replays, searches) in <stats prefix>.rank<r>.json.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P scripts/gpu_cli_multirank.py STATS_PREFIX <cli args...>
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multihop_dense_retrieval_amd import eval_mhop_retrieval  # noqa: E402

if __name__ == "__main__":
    prefix, argv = sys.argv[1], sys.argv[2:]
    eval_mhop_retrieval.main(argv)
    with open(f"{prefix}.rank{os.environ.get('RANK', '0')}.json", "w") as f:
        json.dump(eval_mhop_retrieval.LAST_RUN, f)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()

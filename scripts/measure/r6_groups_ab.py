#!/usr/bin/env python3
"""Round 6: how the passes of a > 256-query search share the queries (MDR_MIPS_EVEN_GROUPS = 0 old cut 256 + ... + rest, 1 even in units of 32-query waves (default),
2 even + groups of <= 128 queries on the 16-queries-per-wave kernels). One process per mode (the knob is read once); HIP events, 10 back-to-back calls.
usage: python scripts/measure/r6_groups_ab.py rows storage nq:k [nq:k ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd import index as mdr_index  # noqa: E402

rows, storage = int(sys.argv[1]), sys.argv[2]
shapes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[3:]]
dev = torch.device("cuda", 0)
idx = mdr_index.IndexFlatIP(768, device=dev, storage=storage) if storage == "bf16" else mdr_index.IndexFlatIP(768, device=dev)
idx.reserve(rows)
g = torch.Generator(device=dev).manual_seed(0)
for lo in range(0, rows, 250_000):
    idx.add(torch.randn((min(250_000, rows - lo), 768), generator=g, device=dev))
probe = torch.randn((1024, 768), generator=torch.Generator(device=dev).manual_seed(0), device=dev)  # == rows 0..1023 of the corpus
mode = os.environ.get("MDR_MIPS_EVEN_GROUPS", "1")
for nq, k in shapes:
    q = torch.randn((nq, 768), generator=g, device=dev)
    q[: nq // 2] = probe[: nq // 2] + 0.05 * q[: nq // 2]  # half planted (bench-like), half pure noise
    for _ in range(3):
        D, I = idx.search_device(q, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        idx.search_device(q, k)
    e1.record()
    torch.cuda.synchronize()
    ok = bool((I[: nq // 2, 0] == torch.arange(nq // 2, device=dev)).all())
    print(f"groups={mode} rows {rows} {storage} nq {nq} k {k} {idx.last_kernel():32s} {e0.elapsed_time(e1) / 10:8.3f} ms  planted-found {ok} checksum {int(I.sum())}", flush=True)

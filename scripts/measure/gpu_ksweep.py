#!/usr/bin/env python3
"""Timing sweep of mdr_index_search over k and kernel variant (HIP events, device-resident queries).
usage: python scripts/measure/gpu_ksweep.py [rows] [storage]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd import index as mdr_index  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
storage = sys.argv[2] if len(sys.argv) > 2 else "f32x2h"
dev = torch.device("cuda", 0)
DIM = int(os.environ.get("SWEEP_DIM", "768"))
idx = mdr_index.IndexFlatIP(DIM, device=dev, storage=storage) if storage == "bf16" else mdr_index.IndexFlatIP(DIM, device=dev)
idx.reserve(rows)
g = torch.Generator(device=dev).manual_seed(0)
for lo in range(0, rows, 250_000):
    idx.add(torch.randn((min(250_000, rows - lo), DIM), generator=g, device=dev))
planted_mode = os.environ.get("SWEEP_PLANTED") == "1"
probe = torch.randn((800, DIM), generator=torch.Generator(device=dev).manual_seed(0), device=dev)  # == rows 0..799 of the corpus
for nq in (100, 800) if not os.environ.get("SWEEP_NQ") else (int(os.environ["SWEEP_NQ"]),):
    q = torch.randn((nq, DIM), generator=g, device=dev)
    if planted_mode:  # bench-like queries: a corpus row plus 5% noise (one clear winner per query)
        q = probe[:nq] + 0.05 * q
    for k in (1, 4, 8, 100) if not os.environ.get("SWEEP_K") else (int(os.environ["SWEEP_K"]),):
        for variant, name in ((3, "screen"), (1 if storage == "bf16" else 2, "exact")):
            if name == "exact" and storage == "bf16" and (rows > 1_000_000 or nq > 100):
                continue  # generic fp32 kernel: too slow to be worth GPU minutes
            idx.set_variant(variant)
            for _ in range(2):
                idx.search_device(q, k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                idx.search_device(q, k)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"rows {rows} {storage} nq {nq} k {k} {name:6s} {idx.last_kernel():34s} {ms:8.3f} ms  "
                  f"{idx.stream_bytes() / ms / 1e6:8.1f} GB/s algorithmic", flush=True)
idx.set_variant(0)

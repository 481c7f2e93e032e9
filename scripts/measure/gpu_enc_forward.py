"""Hop-2-shaped encoder forwards only (100 sequences, lengths U[90,330] padded to 350, random-init RoBERTa-base): the workload for
per-kernel profiles of the encoder without the index (scripts/measure/gpu_attn_ab.sh). ENC_LEN=lo,hi overrides the length range, ENC_L the padded
length (ENC_LEN=8,40 ENC_L=70: the hop-1 shape)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd.retriever import RobertaRetriever  # noqa: E402

lo, hi = (int(v) for v in os.environ.get("ENC_LEN", "90,330").split(","))
B, L = 100, int(os.environ.get("ENC_L", "350"))
torch.manual_seed(0)
m = RobertaRetriever.random_init(device="cuda", seed=3)
lens = torch.randint(lo, hi + 1, (B,))
ids = torch.ones((B, L), dtype=torch.int64)
mask = torch.zeros((B, L), dtype=torch.int64)
for b in range(B):
    n = int(lens[b])
    ids[b, :n] = torch.randint(3, 50000, (n,))
    ids[b, 0], ids[b, n - 1] = 0, 2
    mask[b, :n] = 1
ids, mask = ids.cuda(), mask.cuda()
for _ in range(3):
    m.encode_q(ids, mask, None)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    m.encode_q(ids, mask, None)
torch.cuda.synchronize()
print(f"tokens {int(lens.sum())}  forward {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")

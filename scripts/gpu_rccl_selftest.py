#!/usr/bin/env python3
"""RCCL path on ONE GPU: a one-rank "nccl" process group runs the same collectives the N-GPU bench issues
(all_gather_into_tensor of fp32 embeddings and of the packed int64 (score bits, id) lists) + the merge kernel, and the
weak-scaling bench step, so API misuse on RCCL (dtypes, shapes, stream order, graph capture next to collectives) shows
up without a multi-GPU node. Launch: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 \
--master-port 29533 scripts/gpu_rccl_selftest.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multihop_dense_retrieval_amd import index as mdr_index  # noqa: E402
from multihop_dense_retrieval_amd.retriever import RobertaRetriever  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((100, 768), generator=g, device=dev)
y = mdr_index.all_gather_dim0(x, 1)
assert torch.equal(x, y)
N = 200_000
sidx = mdr_index.ShardedIndexFlatIP(768, N)
sidx.add_local(torch.randn((N, 768), generator=g, device=dev))
q = torch.randn((130, 768), generator=g, device=dev)
for k in (1, 8):
    D, I = sidx.local.search_device(q, k)
    Dm, Im = sidx.search_gathered(D, I, force=True)  # all_gather of the packed int64 buffer + mdr_topk_merge
    Dp, Ip = sidx.search_device(q, k, force=True)    # round 4's exchange: search -> packed block, one RCCL all-gather of uint8 blocks, mdr_topk_merge_packed
    assert torch.equal(Ip, Im) and torch.equal(Dp, Dm), "packed exchange differs from the generic one"
    assert torch.equal(Dm, D) and torch.equal(Im, I), k
    # the faiss-style numpy surface over the same collective: host queries in, host results out (this used to hand
    # CPU tensors to the nccl group and to the HIP merge kernel)
    Dn, In = sidx.search(q.cpu().numpy(), k, force=True)
    assert isinstance(Dn, np.ndarray) and np.array_equal(Dn, D.cpu().numpy()) and np.array_equal(In, I.cpu().numpy()), k
# an encoder graph replay between collectives (the bench's weak-scaling step does exactly this)
enc = RobertaRetriever.random_init(device=dev, seed=3)
ids = torch.randint(3, 50265, (16, 40), generator=g, device=dev)
ids[:, 0] = 0
mask = torch.ones_like(ids)
for _ in range(3):
    e = mdr_index.all_gather_dim0(enc.encode_q(ids, mask, None), 1)
    D, I = sidx.search_gathered(*sidx.local.search_device(e.contiguous(), 1), force=True)
dist.barrier()
torch.cuda.synchronize()
assert bool(torch.isfinite(D).all())
dist.destroy_process_group()
print("rccl selftest ok")

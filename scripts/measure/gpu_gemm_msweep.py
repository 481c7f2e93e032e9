#!/usr/bin/env python3
"""Time of the encoder's four GEMM shapes over the number of rows M, 19 k .. 24 k in steps of one 256-row tile: where the tile-round steps of the
persistent kernels are and what the 128x128 tail does to them (kernel 0 = the shape heuristic the forward uses)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
SHAPES = [("qkv", 2304, 768, 0), ("out-proj", 768, 768, 0), ("ffn1+gelu", 3072, 768, 1), ("ffn2", 768, 3072, 0)]
Ms = [256 * t for t in range(76, 95)]
Mmax = max(Ms)
g = torch.Generator(device=dev).manual_seed(0)
print("rows(tiles)".ljust(14) + "".join(n.rjust(12) for n, _, _, _ in SHAPES) + "   sum us", flush=True)
mats = {}
for name, N, K, epi in SHAPES:
    mats[name] = (torch.randn((Mmax, K), generator=g, device=dev).half(), (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).half(),
                  torch.randn((N,), generator=g, device=dev), torch.empty((Mmax, N), device=dev, dtype=torch.float16))
for M in Ms:
    row, tot = f"{M} ({M // 256})".ljust(14), 0.0
    for name, N, K, epi in SHAPES:
        A, W, b, out = mats[name]
        st = torch.cuda.current_stream().cuda_stream

        def call():
            _lib.check(L.mdr_test_gemm_f16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, None, N, K, out.data_ptr(), epi, 0, 0, st))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        row += f"{us:12.1f}"
    print(row + f"{tot:9.1f}", flush=True)

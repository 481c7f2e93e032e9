#!/usr/bin/env python3
"""Micro-benchmark of the encoder GEMM kernels through the mdr_test_gemm_f16 hook (HIP events, 20 launches each).
usage: python scripts/measure/gpu_gemm_bench.py [M] [kernels...]
Ablations / the timeline are VARIANT BUILDS (python -m multihop_dense_retrieval_amd.build -DMDR_GEMM_ABL=n --out=libmdrhip_abl<n>.so),
selected with MDR_LIB_PATH (scripts/measure/gpu_gemm_ab.sh interleaves several); the product library has no such switch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20611
kernels = [int(a) for a in sys.argv[2:]] or [4, 5]
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
SHAPES = (("qkv", 2304, 768, 0), ("out", 768, 768, 3), ("ffn1", 3072, 768, 1), ("ffn2", 768, 3072, 3))
if os.environ.get("GEMM_SHAPES"):  # name:N:K:epilogue,...
    SHAPES = tuple((a, int(b), int(c), int(d)) for a, b, c, d in (t.split(":") for t in os.environ["GEMM_SHAPES"].split(",")))
for name, N, K, epi in SHAPES:
    A = torch.randn((M, K), generator=g, device="cuda").half()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).half()
    b = torch.randn((N,), generator=g, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)  # large enough for every epilogue
    if os.environ.get("GEMM_TORCH_REF"):  # what the vendor library does on the same shape (reference point only, never in the product path)
        bh = b.half()
        for _ in range(3):
            torch.nn.functional.linear(A, W, bh)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.nn.functional.linear(A, W, bh)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:5s} M={M} N={N} K={K} torch F.linear (hipBLASLt, bias, f16 out, no GELU): {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    # SWEEP_ENV=NAME SWEEP_VALUES=a,b,c: re-time every kernel with the library's measurement knob NAME set to each value
    sweep = [(os.environ["SWEEP_ENV"], v) for v in os.environ.get("SWEEP_VALUES", "").split(",")] if os.environ.get("SWEEP_ENV") else [None]
    for kern, knob in [(k, w) for k in kernels for w in sweep]:
        if knob is not None:
            os.environ[knob[0]] = knob[1]
        for _ in range(3):
            _lib.check(L.mdr_test_gemm_f16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, None, N, K, out.data_ptr(), epi, kern, 0, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.mdr_test_gemm_f16(A.data_ptr(), W.data_ptr(), b.data_ptr(), M, None, N, K, out.data_ptr(), epi, kern, 0, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:5s} M={M} N={N} K={K} kernel {kern}{'' if knob is None else ' ' + knob[0] + '=' + knob[1]}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
        if hasattr(L, "mdr_test_gemm_stamps") and kern in (6, 7):  # a -DMDR_GEMM_ABL=5 build (MDR_LIB_PATH): s_memtime timeline of wave 0, shader cycles per K-tile
            import ctypes
            buf = (ctypes.c_uint64 * 8)()
            _lib.check(L.mdr_test_gemm_stamps(buf, 1))
            kt = max(1, buf[7])
            names = ("waitA+barA", "sp1", "sp2", "sp3", "waitB+barB", "sp4", "epilogue") if kern == 6 else ("top0", "body0", "top1", "body1", "-", "-", "epilogue")
            print("      cycles per K-tile (wave 0 mean): " + "  ".join(f"{n} {buf[i] / kt:7.1f}" for i, n in enumerate(names))
                  + f"   total {sum(buf[:7]) / kt:7.1f}", flush=True)

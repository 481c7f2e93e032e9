#!/usr/bin/env python3
"""Where does a small host->device transfer spend its host time on this box? (pipeline._h2d measured ~1.1 ms + 5 us/KB per call.)"""
import ctypes
import time

import numpy as np
import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int


def t(f, n=50):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    el = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return round(el, 4)


for shape in ((100, 70), (100, 350)):
    arr = np.random.randint(0, 50000, shape).astype(np.int64)
    src = torch.from_numpy(arr)
    pin = torch.empty(shape, dtype=torch.int64, pin_memory=True)
    dst = torch.empty(shape, dtype=torch.int64, device=dev)
    busy = torch.randn((4096, 4096), device=dev)

    def load_gpu():
        for _ in range(20):
            busy @ busy

    res = {}
    res["np->pinned copy_"] = t(lambda: pin.copy_(src))
    res["np->pinned np.copyto"] = t(lambda: np.copyto(pin.numpy(), arr))
    res["pinned.to(dev, nb)"] = t(lambda: pin.to(dev, non_blocking=True))
    res["dst.copy_(pinned, nb)"] = t(lambda: dst.copy_(pin, non_blocking=True))
    res["pageable.to(dev)"] = t(lambda: src.to(dev))
    st = torch.cuda.current_stream().cuda_stream
    res["hipMemcpyAsync(pinned)"] = t(lambda: hip.hipMemcpyAsync(dst.data_ptr(), pin.data_ptr(), arr.nbytes, 1, st))
    res["empty(pin_memory)"] = t(lambda: torch.empty(shape, dtype=torch.int64, pin_memory=True))
    res["is_pinned()"] = t(lambda: pin.is_pinned())
    res["event record+query"] = t(lambda: (lambda e: (e.record(), e.query()))(torch.cuda.Event()))
    # the same with the GPU busy (a queue of matmuls in front)
    def busy_case(f):
        def g():
            load_gpu()
            t0 = time.perf_counter()
            f()
            return time.perf_counter() - t0
        g()
        torch.cuda.synchronize()
        xs = []
        for _ in range(10):
            xs.append(g())
            torch.cuda.synchronize()
        return round(float(np.median(xs)) * 1e3, 4)
    res["BUSY pinned.to(dev, nb)"] = busy_case(lambda: pin.to(dev, non_blocking=True))
    res["BUSY dst.copy_(pinned, nb)"] = busy_case(lambda: dst.copy_(pin, non_blocking=True))
    res["BUSY hipMemcpyAsync(pinned)"] = busy_case(lambda: hip.hipMemcpyAsync(dst.data_ptr(), pin.data_ptr(), arr.nbytes, 1, st))
    res["BUSY pageable.to(dev)"] = busy_case(lambda: src.to(dev))
    host = torch.empty(shape, dtype=torch.int64, pin_memory=True)
    res["BUSY host.copy_(dev, nb) D2H"] = busy_case(lambda: host.copy_(dst, non_blocking=True))
    res["BUSY hipMemcpyAsync D2H"] = busy_case(lambda: hip.hipMemcpyAsync(host.data_ptr(), dst.data_ptr(), arr.nbytes, 2, st))
    print(shape, arr.nbytes, "bytes; ms per call:", res, flush=True)

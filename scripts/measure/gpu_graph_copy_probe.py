#!/usr/bin/env python3
"""Does an async copy enqueued behind a hipGraph launch block the host until the graph has run? (ROCm 7.0 / torch 2.10)"""
import time

import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a = torch.randn((4096, 4096), device=dev)
pin = torch.empty((100, 350), dtype=torch.int64, pin_memory=True)
dst = torch.empty((100, 350), dtype=torch.int64, device=dev)
host = torch.empty((100, 350), dtype=torch.int64, pin_memory=True)


def work():
    x = a
    for _ in range(12):
        x = x @ a * 1e-3
    return x


work()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work()
torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); print("graph alone ms", round((time.perf_counter() - t0) * 1e3, 3))


def case(name, launch, op):
    xs = []
    for _ in range(8):
        torch.cuda.synchronize()
        launch()
        t0 = time.perf_counter()
        op()
        xs.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    print(f"{name:52s} host ms median {sorted(xs)[len(xs) // 2]:.3f}  min {min(xs):.3f}", flush=True)


side = torch.cuda.Stream()
for lname, launch in (("eager kernels", work), ("graph replay", g.replay)):
    case(f"{lname} -> H2D pinned.to(dev, nb)", launch, lambda: pin.to(dev, non_blocking=True))
    case(f"{lname} -> H2D dst.copy_(pinned, nb)", launch, lambda: dst.copy_(pin, non_blocking=True))
    case(f"{lname} -> D2H host.copy_(dev, nb)", launch, lambda: host.copy_(dst, non_blocking=True))
    case(f"{lname} -> small kernel", launch, lambda: dst.add_(1))
    case(f"{lname} -> graph replay again", launch, g.replay)
    case(f"{lname} -> event record", launch, lambda: torch.cuda.Event().record())
    def on_side():
        with torch.cuda.stream(side):
            pin.to(dev, non_blocking=True)
    case(f"{lname} -> H2D on ANOTHER stream", launch, on_side)

#!/usr/bin/env python3
"""Host-side probe for the CPU baseline: usable cores (affinity, cgroup quota) and the oracle's speed vs thread count."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    lib = ctypes.CDLL(so)
    f = lib.mdr_oracle_flat_ip_search
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(0)
    xb = rng.standard_normal((500_000, 768), dtype=np.float32)
    q = rng.standard_normal((100, 768), dtype=np.float32)
    D = np.empty((100, 1), np.float32)
    I = np.empty((100, 1), np.int64)
    f(q.ctypes.data, 100, xb.ctypes.data, xb.shape[0], 768, 1, D.ctypes.data, I.ctypes.data, 0)
    t = time.perf_counter()
    f(q.ctypes.data, 100, xb.ctypes.data, xb.shape[0], 768, 1, D.ctypes.data, I.ctypes.data, 0)
    dt = time.perf_counter() - t
    print(f"threads {os.environ.get('OMP_NUM_THREADS')}: {dt * 1e3:8.1f} ms per 500k-row search -> {2 * 100 * 768 * 5e5 / dt / 1e9:7.1f} GFLOP/s", flush=True)
    t = time.perf_counter()
    S = q @ xb.T
    i = S.argmax(1)
    dt = time.perf_counter() - t
    print(f"   numpy sgemm+argmax: {dt * 1e3:8.1f} ms", flush=True)
    sys.exit(0)
for n in (8, 16, 32, 64, 128):
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, OMP_NUM_THREADS=str(n), OPENBLAS_NUM_THREADS=str(min(n, 64))))

import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests` in the build container
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        p = os.path.join(GOLDEN, name)
        if name.endswith(".json"):
            import json
            with open(p) as f:
                return json.load(f)
        return np.load(p)
    return load


class OracleLib:
    """ctypes view of oracle/liboracle.so (the CPU restatement of FAISS flat-IP search)."""

    def __init__(self):
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        src = os.path.join(ROOT, "oracle", "flat_ip_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.lib = ctypes.CDLL(so)
        f = self.lib.mdr_oracle_flat_ip_search
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]

    def search(self, x, xb, k, nthreads=0):
        x = np.ascontiguousarray(x, np.float32)
        xb = np.ascontiguousarray(xb, np.float32)
        nq, d = x.shape
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        rc = self.lib.mdr_oracle_flat_ip_search(x.ctypes.data, nq, xb.ctypes.data, xb.shape[0], d, k, D.ctypes.data, I.ctypes.data,
                                                nthreads)
        assert rc == 0
        return D, I


@pytest.fixture(scope="session")
def oracle():
    return OracleLib()

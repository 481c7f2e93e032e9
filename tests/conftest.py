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


@pytest.fixture(scope="session")
def tiny_roberta_tokenizer(tmp_path_factory):
    """A REAL HuggingFace RobertaTokenizer (byte-level BPE) over the tiny vocabulary of tests/golden/tiny_bpe (see
    make_tiny_bpe.py; the roberta-base files do not exist offline), saved as a model directory and loaded back through
    AutoTokenizer.from_pretrained(<local dir>) -- the CLIs' own loading path."""
    import json
    transformers = pytest.importorskip("transformers")
    bpe = os.path.join(GOLDEN, "tiny_bpe")
    with open(os.path.join(bpe, "vocab.json")) as f:
        vocab = json.load(f)
    with open(os.path.join(bpe, "merges.txt")) as f:
        merges = [tuple(ln.split()) for ln in f.read().split("\n") if ln and not ln.startswith("#")]
    t = transformers.RobertaTokenizer(vocab=vocab, merges=merges)
    d = tmp_path_factory.mktemp("tiny_roberta")
    t.save_pretrained(str(d))
    t2 = transformers.AutoTokenizer.from_pretrained(str(d))
    assert "Roberta" in t2.__class__.__name__ and (t2.bos_token_id, t2.pad_token_id, t2.eos_token_id) == (0, 1, 2)
    return t2

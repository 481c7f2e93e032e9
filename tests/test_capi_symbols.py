"""CPU (-m "not gpu"): libmdrhip.so builds for gfx950 without a GPU, loads, and exports every entry point that
include/mdr_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from multihop_dense_retrieval_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_library_builds_loads_and_exports_everything():
    from multihop_dense_retrieval_amd import build
    path = build.build_lib()
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/mdr_hip.h but not exported"
    lib.mdr_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.mdr_version()


def test_product_path_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from multihop_dense_retrieval_amd import index, retriever
    with pytest.raises(RuntimeError):
        index.IndexFlatIP(768)
    with pytest.raises(RuntimeError):
        retriever.RobertaRetriever(retriever.RobertaConfig(), None).to("cpu")

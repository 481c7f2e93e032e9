"""CPU (-m "not gpu"): libmdrhip.so builds for gfx950 without a GPU, loads, and exports every entry point that
include/mdr_hip.h declares (no compute calls here)."""
import ctypes
import os
import re
import sys

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


def test_product_library_holds_no_measurement_code():
    """switches that make a kernel return wrong numbers (ablations) or that record timelines are COMPILE-TIME
    macros of variant builds. The product library therefore (a) exports none of include/mdr_hip_measure.h, (b) does not contain the
    names of those switches as strings (nothing reads them from the environment), and (c) the only environment variables the
    native sources read are the ones below, each of which selects between kernels that return the same results."""
    import glob
    from multihop_dense_retrieval_amd import build
    path = build.build_lib()
    lib = ctypes.CDLL(path)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mdr_hip_measure.h")).read(), flags=re.S)
    hooks = sorted(set(re.findall(r"\b(mdr_[a-z0-9_]+)\s*\(", text)))
    assert hooks == ["mdr_stream_create_cu_range", "mdr_stream_destroy", "mdr_test_attn_stamps", "mdr_test_gemm_stamps", "mdr_test_i8_stamps"]
    for name in hooks:
        assert not hasattr(lib, name), f"{name} is a measurement hook but the product library exports it"
    assert len([n for n in declared_symbols()]) == 27  # round 6: the CU-lane pair moved to the measurement header
    blob = open(path, "rb").read()
    for knob in (b"MDR_GEMM_ABL", b"MDR_GEMM_EPI", b"MDR_I8_ABL", b"MDR_ATTN_ABL", b"g_gemm_stamp", b"g_i8_stamp", b"g_attn_stamp", b"g_stream_cus", b"mips_gemmk_kernel"):
        assert knob not in blob, f"{knob!r} found in the product library"
    # (MDR_UPLOAD_THREADS: memcpy threads of the host-upload pipeline; MDR_MIPS_EVEN_GROUPS: how the passes of a > 256-query call share the queries;
    #  MDR_MIPS_I8_CB: forces the int8 tier's query split on / off -- any lambda is correct, same ids and scores; MDR_MIPS_GEMMK is read by -DMDR_MIPS_GEMMK=1
    #  measurement builds only)
    allowed = {"MDR_GEMM_CFG", "MDR_MIPS_WIDE", "MDR_MIPS_I8", "MDR_MIPS_GEMMK", "MDR_UPLOAD_THREADS", "MDR_MIPS_EVEN_GROUPS", "MDR_MIPS_I8_CB"}
    seen = set()
    for src in glob.glob(os.path.join(ROOT, "multihop_dense_retrieval_amd", "csrc", "*")):
        seen |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(src).read()))
    assert seen <= allowed, f"environment variables read by the native sources: {sorted(seen)}; result-neutral ones allowed: {sorted(allowed)}"


def test_product_path_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from multihop_dense_retrieval_amd import index, retriever
    with pytest.raises(RuntimeError):
        index.IndexFlatIP(768)
    with pytest.raises(RuntimeError):
        retriever.RobertaRetriever(retriever.RobertaConfig(), None).to("cpu")


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it.
    Walk every module of the package (and the drop-in entry points under scripts/) and fail on any import of it, static or
    dynamic (importlib / __import__ with the literal name), or any dlopen of oracle/liboracle.so."""
    import ast
    import glob
    pkg = os.path.join(ROOT, "multihop_dense_retrieval_amd")
    files = sorted(glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True)) + [
        os.path.join(ROOT, "scripts", "encode_corpus.py")] + sorted(glob.glob(os.path.join(ROOT, "scripts", "eval", "*.py")))
    assert len(files) > 10
    for path in files:
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            elif isinstance(node, ast.Constant) and isinstance(node.value, str) and node is not ast.get_docstring:
                # string constants that would name the checker in a dynamic import / dlopen
                names = [node.value] if re.fullmatch(r"oracle(\.\w+)*|.*liboracle\.so", node.value) else []
            else:
                continue
            for n in names:
                assert not (n == "oracle" or n.startswith("oracle.") or n.endswith("liboracle.so")), f"{path} reaches into oracle/: {n}"
    # the native sources must not link or include it either
    for path in glob.glob(os.path.join(pkg, "csrc", "*")):
        assert "oracle" not in open(path).read(), path


def test_docs_cite_tests_and_files_that_exist():  # (NEGATIVE_RESULTS.md is history: its citations are not kept alive)
    """DESIGN.md / INTEGRATION.md name tests (`tests/x.py::test_y`) and files (`profiles/...`, `scripts/...`) as evidence: every
    such reference must resolve, so the documents cannot drift from the tree (VERDICT r1 found a dangling test name)."""
    import glob
    missing = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for path, name in re.findall(r"`(tests/[\w/]+\.py)::(\w+)", text):
            src = open(os.path.join(ROOT, path)).read() if os.path.exists(os.path.join(ROOT, path)) else ""
            if not re.search(rf"def {re.escape(name)}\w*\(", src):  # (the documents abbreviate long names with an ellipsis: prefix match)
                missing.append(f"{doc}: {path}::{name}")
        for path in re.findall(r"`((?:tests|scripts/gpu_|scripts/measure/|profiles|oracle|include)[\w./-]+?\.(?:py|sh|txt|csv|json|h|c|md))[`:]", text):
            if "*" in path or "<" in path:
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append(f"{doc}: {path}")
    assert not missing, missing


def test_hand_scheduled_gemm_keeps_its_accumulators_to_itself():
    """gemm_quad_kernel leaves its accumulators in a[0:255] across inline-asm statements; the compiler must not touch an AGPR or spill in that
    kernel (scripts/check_quad_agprs.py compiles csrc/mdr_encoder.hip to assembly and reads every instantiation)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_quad_agprs.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


def test_generated_k_loop_is_what_the_generator_writes():
    """csrc/mdr_encoder_gemm_quad_loop.inc is generated; the committed file must be the generator's output."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_gemm_quad_asm.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    assert r.stdout == open(os.path.join(ROOT, "multihop_dense_retrieval_amd", "csrc", "mdr_encoder_gemm_quad_loop.inc")).read()

"""CPU: the measurement recipes that used to be product flags (scripts/measure/bench_loops.py, loop_variants.py, cu_lanes.py: VERDICT r5 item 7) still parse and still
fit the product classes they extend -- they are not exercised by any other test, and a silent rot would make the recorded negatives unreproducible."""
import ast
import glob
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEASURE = os.path.join(ROOT, "scripts", "measure")


def test_every_measurement_script_parses():
    files = sorted(glob.glob(os.path.join(MEASURE, "*.py"))) + [os.path.join(ROOT, "scripts", "structured_corpora.py")]
    assert len(files) > 10
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_loop_variants_extend_the_product_pipeline():
    pytest.importorskip("torch")
    sys.path.insert(0, MEASURE)
    try:
        lv = importlib.import_module("loop_variants")
    finally:
        sys.path.remove(MEASURE)
    from multihop_dense_retrieval_amd import mhop
    assert issubclass(lv.VariantTwoHop, mhop.SyntheticTwoHop)
    for name in ("_step_deep", "_step_shift", "_step_pipelined_grouped", "_hop1_of"):
        assert callable(getattr(lv.VariantTwoHop, name)), name
    # what the variants call on the base class is still there
    for name in ("_encode", "_search", "_hop1_only", "_hop2_inputs", "_interleave", "_own", "_mark", "_nxt", "step"):
        assert callable(getattr(mhop.SyntheticTwoHop, name)), name
    # the product class itself no longer carries them
    assert not hasattr(mhop.SyntheticTwoHop, "_step_deep") and not hasattr(mhop.SyntheticTwoHop, "_step_shift")


def test_bench_has_no_variant_flags_and_the_wrapper_owns_them():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--loop", "--hop1-group", "--lane-cus"):
        assert f'"{flag}"' not in src, flag
    wrapper = open(os.path.join(MEASURE, "bench_loops.py")).read()
    for flag in ("--loop", "--hop1-group", "--lane-cus"):
        assert f'"{flag}"' in wrapper, flag

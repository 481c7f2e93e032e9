"""GPU parity tests (-m gpu) of the int8 screening tier (k = 1, F32X2H storage, d = 768; csrc/mdr_mips.hip, mips_screen8*_kernel):
whatever the tier does -- decide alone, hand over to the fp16 screen, hand over twice -- ids and scores are those of the exact
stream kernel, and the telemetry hook says which way it went. Reference: faiss.IndexFlatIP.search as called at
/root/reference/scripts/eval/eval_mhop_retrieval.py:155,179 (exact inner products, ties to the lowest id)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
D_ = 768


@pytest.fixture(scope="module")
def mdr():
    from multihop_dense_retrieval_amd import _lib, index
    _lib.lib()
    return index


def exact(idx, q):
    idx.set_variant(2)
    try:
        return idx.search(q, 1)
    finally:
        idx.set_variant(0)


def check(idx, q, *, expect_i8_decides=None):
    De, Ie = exact(idx, q)
    for v in (0, 4):  # int8 tier in front / fp16 screen alone
        idx.set_variant(v)
        D, I = idx.search(q, 1)
        t = idx.telemetry(q.shape[0], 1)
        idx.set_variant(0)
        assert torch.equal(I, Ie), v
        assert torch.equal(D, De) or float((D - De).abs().max()) <= 2e-6 * float(De.abs().max()) + 1e-30, v
        assert t["i8_tier"] == (v == 0)
        if v == 0 and expect_i8_decides is not None:
            assert (not t["i8_overflow"]) == expect_i8_decides, t
    return t


@pytest.mark.parametrize("n", [1, 31, 32, 33, 95, 4097])
@pytest.mark.parametrize("nq", [1, 100, 130])
def test_small_and_ragged_indexes(mdr, n, nq):
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + nq)
    idx = mdr.IndexFlatIP(D_)
    idx.add(torch.randn((n, D_), generator=g, device="cuda"))
    check(idx, torch.randn((nq, D_), generator=g, device="cuda"))


def test_no_clear_winner_queries_are_decided_by_the_tier(mdr):
    """iid queries against iid rows: hundreds of rows per query sit inside the int8 band; most candidates are emitted early against a
    loose bound and dropped before re-scoring."""
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = mdr.IndexFlatIP(D_)
    idx.reserve(600_000)
    for _ in range(3):  # several add() calls: the plane and its bound constants accumulate
        idx.add(torch.randn((200_000, D_), generator=g, device="cuda"))
    for nq in (100, 200, 300, 800):  # 16 queries per wave / 32 per wave / two and four groups of 256 (a shard's load at N = 8, weak scaling)
        q = torch.randn((nq, D_), generator=g, device="cuda")
        idx.set_variant(0)
        idx.search(q, 1)
        t = idx.telemetry(nq, 1)
        assert t["i8_tier"] and not t["i8_overflow"] and t["fallback"] == 0
        groups = -(-nq // 256)  # `candidates` counts what the LAST group of 256 emitted, `i8_refined` what all groups re-scored
        assert 0 < t["i8_refined"] <= t["candidates"] * 2 * groups + 1
        assert ("mips_screen8w_kernel" if nq > 128 else "mips_screen8_kernel") in idx.last_kernel()
        check(idx, q, expect_i8_decides=True)


def test_growth_keeps_the_int8_plane(mdr):
    g = torch.Generator(device="cuda").manual_seed(5)
    idx = mdr.IndexFlatIP(D_)  # no reserve: the planes are re-allocated and copied as the index grows
    rows = []
    for n in (1000, 37, 5000, 64):
        x = torch.randn((n, D_), generator=g, device="cuda")
        idx.add(x)
        rows.append(x)
        q = torch.cat(rows)[-40:] + 0.02 * torch.randn((40, D_), generator=g, device="cuda")
        check(idx, q, expect_i8_decides=True)


def test_rows_and_queries_of_very_different_scales(mdr):
    """Per-row scales: rows spanning 9 orders of magnitude (F32X2H storage takes |x| <= 32768), an all-zero row, an all-zero query, a query with one huge element."""
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((20_000, D_), generator=g, device="cuda")
    x *= torch.logspace(-6, 3, 20_000, device="cuda")[torch.randperm(20_000, generator=g, device="cuda")][:, None]
    x[123] = 0
    idx = mdr.IndexFlatIP(D_)
    idx.add(x)
    q = torch.randn((64, D_), generator=g, device="cuda")
    q[5] = 0
    q[6, 17] = 3e4
    q[7] *= 1e-20
    q[8] *= 1e20
    check(idx, q)


def test_loose_bound_hands_over_to_the_fp16_screen(mdr):
    """Heavy-tailed rows (one element carries the row: its scale wipes out the rest) and rows with a large common mean make the int8
    bound wide: thousands of survivors per query -> the tier declares itself overflowed and the fp16 screen decides."""
    g = torch.Generator(device="cuda").manual_seed(9)
    n = 100_000
    x = torch.randn((n, D_), generator=g, device="cuda")
    x[torch.arange(n, device="cuda"), torch.randint(0, D_, (n,), generator=g, device="cuda")] = 400.0
    idx = mdr.IndexFlatIP(D_)
    idx.add(x)
    q = torch.randn((100, D_), generator=g, device="cuda")
    t = check(idx, q)
    idx.set_variant(0)
    idx.search(q, 1)
    t = idx.telemetry(100, 1)
    assert t["i8_tier"] and t["i8_overflow"], t
    assert t["fallback"] == 0, t  # ... and the fp16 screen behind it needs no exact pass


def test_env_switch_leaves_the_plane_out(mdr, monkeypatch):
    """MDR_MIPS_I8=0 is read once per process, so this only checks the variant-4 spelling of the same thing."""
    g = torch.Generator(device="cuda").manual_seed(11)
    idx = mdr.IndexFlatIP(D_)
    idx.add(torch.randn((5000, D_), generator=g, device="cuda"))
    q = torch.randn((10, D_), generator=g, device="cuda")
    idx.set_variant(4)
    idx.search(q, 1)
    assert "mips_screen_kernel" in idx.last_kernel() and not idx.telemetry(10, 1)["i8_tier"]
    idx.set_variant(0)
    idx.search(q, 1)
    assert "mips_screen8_kernel" in idx.last_kernel()


def test_compact_storage_is_the_same_index_without_the_plane(mdr):
    """MDR_STORE_F32X2H_COMPACT: FAISS's 4 bytes per element, no int8 screening copy; the ids of the default storage, scores from the same exact re-scoring."""
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn((40000, D_), generator=g, device="cuda")
    q = torch.randn((100, D_), generator=g, device="cuda")
    a, b = mdr.IndexFlatIP(D_), mdr.IndexFlatIP(D_, storage="compact")
    a.add(x)
    b.add(x)
    for k in (1, 4):
        Da, Ia = a.search(q, k)
        Db, Ib = b.search(q, k)
        assert torch.equal(torch.as_tensor(Ia), torch.as_tensor(Ib))
        Da, Db = torch.as_tensor(Da), torch.as_tensor(Db)
        assert torch.equal(Da, Db) or float((Da - Db).abs().max()) <= 2e-6 * float(Da.abs().max())
    b.search(q, 1)
    assert "mips_screen_kernel" in b.last_kernel() and not b.telemetry(100, 1)["i8_tier"]
    a.search(q, 1)
    assert a.telemetry(100, 1)["i8_tier"]
    free0 = torch.cuda.mem_get_info()[0]
    c = mdr.IndexFlatIP(D_, storage="compact")
    c.reserve(1_000_000)
    used_compact = free0 - torch.cuda.mem_get_info()[0]
    d = mdr.IndexFlatIP(D_)
    d.reserve(1_000_000)
    used_default = free0 - used_compact - torch.cuda.mem_get_info()[0]
    assert used_compact <= 1_000_000 * D_ * 4 * 1.02 and used_default >= 1_000_000 * D_ * 5 * 0.98


@pytest.mark.parametrize("seed", range(12))
def test_randomised_differential_against_the_exact_kernel(mdr, seed):
    """Random sizes, query counts and data families (gaussian, uniform, sparse, low-rank + noise, clustered with exact duplicates and
    near-duplicates, mixed scales): the k = 1 answer of the tiered search is the exact kernel's, whichever tier decides."""
    rng = np.random.RandomState(1000 + seed)
    g = torch.Generator(device="cuda").manual_seed(2000 + seed)
    n = int(rng.choice([257, 4096, 33_333, 120_001, 300_000]))
    nq = int(rng.choice([1, 17, 100, 128, 129, 200, 256, 257, 300]))
    fam = ["gauss", "uniform", "sparse", "lowrank", "clusters", "scales"][seed % 6]
    x = torch.randn((n, D_), generator=g, device="cuda")
    if fam == "uniform":
        x = torch.rand((n, D_), generator=g, device="cuda") - 0.3
    elif fam == "sparse":
        x = x * (torch.rand((n, D_), generator=g, device="cuda") < 0.05)
    elif fam == "lowrank":
        x = torch.randn((n, 8), generator=g, device="cuda") @ torch.randn((8, D_), generator=g, device="cuda") + 0.05 * x
    elif fam == "clusters":
        c = torch.randn((50, D_), generator=g, device="cuda")
        x = c[torch.randint(0, 50, (n,), generator=g, device="cuda")] + 0.01 * x
        x[n // 2] = x[3]                      # an exact duplicate: lowest id wins
        x[n // 3] = x[3] * (1 - 1e-5)         # and a near-duplicate
    elif fam == "scales":
        x = x * torch.exp(4 * torch.randn((n, 1), generator=g, device="cuda"))
        x = x.clamp(-3e4, 3e4)
    idx = mdr.IndexFlatIP(D_)
    idx.add(x.contiguous())
    q = torch.randn((nq, D_), generator=g, device="cuda")
    if fam in ("clusters", "lowrank"):
        q[: min(nq, 8)] = x[torch.randint(0, n, (min(nq, 8),), generator=g, device="cuda")] + 0.001 * q[: min(nq, 8)]
        q[0] = x[3]
    check(idx, q.contiguous())


def test_more_than_128_queries_with_the_wide_kernels_switched_off():
    """with MDR_MIPS_WIDE=0 (measurement knob: every call stays on the 128-queries-per-pass kernels) a k = 1 call with more
    than 128 queries used to reach run_screen8, which serves ONE group of 128: queries 128.. came back as -1. The int8 tier is now
    only planned for nq <= 128 or with the wide kernels; otherwise the fp16 screen, which loops over groups, decides. The variable is
    read once per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import torch
from multihop_dense_retrieval_amd import index as mi
g = torch.Generator(device='cuda').manual_seed(5)
xb = torch.randn((60000, 768), generator=g, device='cuda')
q = torch.randn((300, 768), generator=g, device='cuda')
q[:150] = xb[torch.arange(150, device='cuda') * 397] + 0.05 * q[:150]
idx = mi.IndexFlatIP(768)
idx.add(xb)
D, I = idx.search_device(q, 1)
kern = idx.last_kernel()
idx.set_variant(2)
De, Ie = idx.search_device(q, 1)
assert torch.equal(I, Ie) and bool((I >= 0).all()), int((I != Ie).sum())
assert float((D - De).abs().max()) <= 1e-3
assert torch.equal(I[:150, 0], torch.arange(150, device='cuda') * 397)
print('ok', kern)
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MDR_MIPS_WIDE="0"), capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "ok mips_screen_kernel<24,1>" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("nq", [100, 200])
def test_query_split_on_rows_with_a_large_common_component(mdr, nq):
    """Round 6: rows AND queries that share a large common vector (what LayerNorm outputs of one trained encoder look like: |c| = 3 x the spread around it). The
    index decides for the query split with its centre (telemetry bit `i8_query_split`), the int8 tier alone decides every query, and the answer is the exact
    kernel's -- for queries of the corpus' kind, for queries WITHOUT the common component (lambda ~ 0: the split must not hurt them) and for queries pointing
    AWAY from it (negative lambda). An i.i.d. corpus keeps the split off."""
    g = torch.Generator(device="cuda").manual_seed(77)
    n = 200_000
    c = 1.0 * torch.randn((D_,), generator=g, device="cuda")           # |c| ~ 27.7
    x = c + 0.33 * torch.randn((n, D_), generator=g, device="cuda")     # spread ~ 9.1
    idx = mdr.IndexFlatIP(D_)
    idx.add(x)
    noise = torch.randn((nq, D_), generator=g, device="cuda")
    kinds = {"same kind": c + 0.33 * noise, "planted": x[torch.arange(nq, device="cuda") * 997] + 0.02 * noise, "no common part": 0.33 * noise,
             "opposite": -c + 0.33 * noise}
    for name, q in kinds.items():
        t = check(idx, q.contiguous(), expect_i8_decides=True)
        idx.search(q.contiguous(), 1)
        t = idx.telemetry(nq, 1)
        assert t["i8_query_split"] and t["i8_tier"] and not t["i8_overflow"] and t["fallback"] == 0, (name, t)
        print(f"query split, nq {nq}, {name}: candidates emitted {t['candidates']}, re-scored {t['i8_refined']}")
    iid = mdr.IndexFlatIP(D_)
    iid.add(torch.randn((50_000, D_), generator=g, device="cuda"))
    iid.search(noise.contiguous(), 1)
    assert not iid.telemetry(nq, 1)["i8_query_split"]


def test_query_split_on_the_hip_encoders_own_outputs(mdr):
    """Rows = this repo's RobertaCtxEncoder outputs for 150 k synthetic passages (random-init roberta-base geometry: a shared LayerNorm bias, nearly collapsed rows --
    the corpus of bench.py's `structured.encoder_geometry`, scripts/structured_corpora.py), queries from the same encoder. The index switches the query split on,
    the int8 tier alone decides, and every id / score is the exact kernel's (round 5's code handed such a corpus over to the slower tiers at 5 M rows)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import structured_corpora as sc
    from multihop_dense_retrieval_amd.retriever import RobertaCtxEncoder
    dev = torch.device("cuda")
    model = RobertaCtxEncoder.random_init(device=dev, seed=3)
    idx = mdr.IndexFlatIP(D_)
    n = 150_000
    idx.reserve(n)
    first = None
    for _, x in sc.encoder_rows(model, n, 8, 24, dev):
        first = x[:50_000].clone() if first is None else first
        idx.add(x)
    assert idx.ntotal == n
    c = first.mean(0)
    assert float(c.norm()) > 2.0 * float((first - c).norm(dim=1).mean())  # the geometry the split is for: the common component dominates
    for nq in (100, 200):
        q = sc.encoder_queries(model, nq, 8, 24, dev)
        t = check(idx, q, expect_i8_decides=True)
        idx.search(q, 1)
        t = idx.telemetry(nq, 1)
        assert t["i8_query_split"] and t["i8_tier"] and not t["i8_overflow"] and t["fallback"] == 0, t
        print(f"encoder-geometry rows, nq {nq}: candidates emitted {t['candidates']}, re-scored {t['i8_refined']}")

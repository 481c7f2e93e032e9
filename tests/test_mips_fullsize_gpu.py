"""GPU parity tests (-m gpu) at BASELINE.json's FULL sizes and on adversarial data, through the C ABI:

  * configs[2] shape: 5 M x 768, k = 1 and k = 4, both storages, against a brute-force pass over every row done with a
    plain torch matmul chunk by chunk (an independent implementation; the CPU oracle needs minutes at this size) and
    against planted known answers;
  * configs[4] single-shard shape: bf16 storage, nq = 800 (beam 8 x 100 questions), k = 8 / 64, against the generic fp32
    kernel and the CPU oracle on the bf16-rounded rows;
  * the hi-plane screen's bound on data it was not tuned on: a large common mean (real dense-retrieval embeddings are
    anisotropic), rows in fp16's subnormal range, queries far outside fp16's range -- with the telemetry hook asserting WHICH
    path produced the answer (screen vs exact fallback).

Tolerances: BASELINE.json north_star -- identical ids (where the runner-up is further away than the scoring noise),
scores within 1e-3 (fp32-accurate storage) / 1e-2 (bf16 storage; exact w.r.t. the rounded rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CHUNK = 250_000
D_ = 768


@pytest.fixture(scope="module")
def mdr():
    from multihop_dense_retrieval_amd import _lib, index
    _lib.lib()
    return index


def chunk(c, rows=CHUNK, seed=77):
    g = torch.Generator(device="cuda").manual_seed(seed * 1_000_003 + c)
    return torch.randn((rows, D_), generator=g, device="cuda")


def brute_force(q, n_chunks, k, transform=None):
    """Top-k of q against all chunks with fp32 matmuls (fp64 for the final candidates): ids, fp64 scores."""
    run_s = torch.full((q.shape[0], k), -float("inf"), device="cuda")
    run_i = torch.full((q.shape[0], k), -1, dtype=torch.int64, device="cuda")
    for c in range(n_chunks):
        blk = chunk(c)
        if transform is not None:
            blk = transform(blk)
        s, i = torch.topk(q @ blk.T, k, dim=1)
        cat_s, cat_i = torch.cat([run_s, s], 1), torch.cat([run_i, i + c * CHUNK], 1)
        run_s, o = torch.topk(cat_s, k, dim=1)
        run_i = torch.gather(cat_i, 1, o)
        del blk
    return run_s, run_i


def exact_scores(q, I, transform=None):
    """fp64 inner products of each query with the rows it was handed back."""
    out = torch.zeros(I.shape, dtype=torch.float64, device="cuda")
    for c in torch.unique(torch.div(I, CHUNK, rounding_mode="floor")).tolist():
        blk = chunk(int(c))
        if transform is not None:
            blk = transform(blk)
        sel = torch.div(I, CHUNK, rounding_mode="floor") == c
        qi, kj = sel.nonzero(as_tuple=True)
        out[qi, kj] = (blk[I[qi, kj] - c * CHUNK].double() * q[qi].double()).sum(1)
        del blk
    return out


def bf16_round(x):
    return x.to(torch.bfloat16).float()


@pytest.fixture(scope="module")
def five_million(mdr):
    """Both storages of the SAME 5 M x 768 corpus (chunk-seeded), built once for the module."""
    built = {}
    for storage in ("f32", "bf16"):
        idx = mdr.IndexFlatIP(D_, storage="bf16") if storage == "bf16" else mdr.IndexFlatIP(D_)
        idx.reserve(20 * CHUNK)
        for c in range(20):
            idx.add(chunk(c))
        assert idx.ntotal == 5_000_000
        built[storage] = idx
    return built


@pytest.mark.parametrize("storage,tol", [("f32", 1e-3), ("bf16", 1e-2)])
@pytest.mark.parametrize("k", [1, 4])
def test_five_million_rows_planted_and_brute_force(five_million, storage, tol, k):
    idx = five_million[storage]
    nq = 100
    g = torch.Generator(device="cuda").manual_seed(5)
    planted = (torch.arange(nq, device="cuda") * 48_611 + 17) % 5_000_000
    q = 0.05 * torch.randn((nq, D_), generator=g, device="cuda")
    rows = torch.stack([chunk(int(p) // CHUNK)[int(p) % CHUNK] for p in planted[:50].tolist()])
    q[:50] += rows  # first half: one clear winner each (known without any reference); second half: pure noise queries
    q[50:] *= 20.0
    D, I = idx.search_device(q.contiguous(), k)
    tr = bf16_round if storage == "bf16" else None
    assert torch.equal(I[:50, 0], planted[:50])
    ex = exact_scores(q, I, tr)
    assert float((ex - D.double()).abs().max()) <= (1e-3 if storage == "f32" else 2e-3)  # exact w.r.t. the STORED rows in both storages
    bs, bi = brute_force(q, 20, k, tr)
    # no row anywhere beats what was returned (fp32 matmul noise 2e-3), and ids agree wherever the decision is not inside that noise
    assert float((bs[:, k - 1] - D[:, k - 1]).max()) <= 2e-3
    differ = (bi != I)
    assert bool(((bs - D).abs()[differ] <= 2e-3).all())
    assert float(differ.float().mean()) <= 0.02
    # against the fp32 rows the bf16 index stays inside the north star's 1e-2
    if storage == "bf16":
        assert float((exact_scores(q, I, None) - D.double()).abs().max() / max(1.0, float(D.abs().max()))) <= tol
    t = idx.telemetry(nq, k)
    assert t["path"] == 3 and t["fallback"] == 0 and 0 < t["candidates"] < 2_000_000, t  # decided by the hi-plane screen alone


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("nq,k", [(100, 100), (100, 250), (300, 250)])
def test_five_million_rows_deep_lists_on_the_screen_path(five_million, storage, nq, k):
    """The reference's downstream runs use beam sizes of 50 / 100 / 250 (README.md:240-241, mdr/qa/train.md:84). k up to 256 stays on the screen-k
    kernels at 5 M rows -- the sample pass keeps one maximum per (workgroup, stage), so its k-th largest is a bound only a few thousand rows deep --
    and returns what a brute-force fp32 matmul over all rows returns."""
    idx = five_million[storage]
    g = torch.Generator(device="cuda").manual_seed(50 + k)
    q = torch.randn((nq, D_), generator=g, device="cuda")
    planted = (torch.arange(nq, device="cuda") * 48_611 + 17) % 5_000_000
    rows = torch.stack([chunk(int(p) // CHUNK)[int(p) % CHUNK] for p in planted[:20].tolist()])
    q[:20] = rows + 0.05 * q[:20]
    D, I = idx.search_device(q.contiguous(), k)
    assert ("mips_screenk32_kernel" if nq > 128 else "mips_screenk_kernel") in idx.last_kernel()
    t = idx.telemetry(nq, k)
    assert t["path"] == 3 and t["fallback"] == 0, t  # decided by the screen, not by the exact pass behind it
    print(f"screen-k at 5 M rows, {storage}, nq {nq} k {k}: candidates per query {t['candidates'] / nq:.0f}")
    tr = bf16_round if storage == "bf16" else None
    assert torch.equal(I[:20, 0], planted[:20])
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and bool((I >= 0).all())
    ex = exact_scores(q, I, tr)
    assert float((ex - D.double()).abs().max()) <= (1e-3 if storage == "f32" else 2e-3)
    bs, bi = brute_force(q, 20, k, tr)
    assert float((bs[:, k - 1] - D[:, k - 1]).max()) <= 2e-3  # nothing anywhere beats the returned k-th
    differ = (bi != I)
    assert bool(((bs - D).abs()[differ] <= 2e-3).all())
    assert float(differ.float().mean()) <= 0.05  # deep ranks sit inside the fp32-matmul noise of each other more often


@pytest.mark.parametrize("k", [8, 64])
def test_bf16_nq800_against_generic_kernel_and_oracle(mdr, oracle, k):
    """BASELINE configs[4], one shard's view: bf16 rows, 800 queries (beam 8 x 100 questions) in four passes of 256 (32 queries per wave)."""
    n = 200_000
    xb = chunk(0, rows=n, seed=901)
    idx = mdr.IndexFlatIP(D_, storage="bf16")
    idx.add(xb)
    g = torch.Generator(device="cuda").manual_seed(9)
    q = torch.randn((800, D_), generator=g, device="cuda")
    q[::7] = xb[torch.arange(0, 800, 7, device="cuda") * 13] + 0.1 * q[::7]
    D, I = idx.search_device(q, k)
    assert "mips_screenk32_kernel" in idx.last_kernel()  # 800 queries: 256 per corpus pass
    idx.set_variant(1)
    try:
        Dg, Ig = idx.search_device(q, k)
    finally:
        idx.set_variant(0)
    noise = 4e-6 * torch.clamp(D.abs(), min=1.0)  # two exact-fp32 kernels, two summation orders
    assert bool(((D - Dg).abs() <= noise).all())
    assert bool(((I == Ig) | ((D - Dg).abs() <= noise)).all()) and float((I == Ig).float().mean()) > 0.999
    # CPU oracle on the rounded rows for a slice of the queries (it needs ~1 s per 100 queries here)
    xr = bf16_round(xb).cpu().numpy()
    Do, Io = oracle.search(q[:96].cpu().numpy(), xr, k)
    Dn, In = D[:96].cpu().numpy(), I[:96].cpu().numpy()
    assert np.abs(Dn - Do).max() <= 1e-3
    same = In == Io
    gap_ok = np.abs(Dn - Do) <= 1e-3
    assert (same | gap_ok).all() and same.mean() > 0.995
    assert idx.telemetry(800, k)["fallback"] == 0


def _oracle_check(oracle, idx, xb_np, q_np, k, score_tol=1e-3):
    D, I = idx.search(q_np, k)
    Do, Io = oracle.search(q_np, xb_np, k)
    scale = max(1.0, float(np.abs(Do).max()))
    assert np.abs(D - Do).max() <= score_tol * scale, np.abs(D - Do).max()
    # ids: identical unless the float64 scores of the two candidates are within the fp32 scoring noise
    x64, q64 = xb_np.astype(np.float64), q_np.astype(np.float64)
    for qi, kj in zip(*np.nonzero(I != Io)):
        a, b = float(x64[I[qi, kj]] @ q64[qi]), float(x64[Io[qi, kj]] @ q64[qi])
        assert abs(a - b) <= 2e-6 * scale * 768 ** 0.5, (qi, kj, a, b)
    return D, I


def test_screen_bound_on_anisotropic_rows_with_a_large_common_mean(mdr, oracle):
    """Rows = a common mean + unit-norm noise (dense-retrieval embeddings are anisotropic like this). The screen's band is
    2B = 2.4e-3 |q| max|x|, the spread of the scores over the rows only |q| / sqrt(d):
      mean norm 200: EVERY row is inside the band -> the candidate lists overflow -> the exact pass must take over;
      mean norm 20 : the band holds a few hundred rows per query -> still decided by the screen + exact re-scoring;
      mean norm 0.5: a handful of candidates.
    The answer is the exact one in all three; the telemetry hook says which path produced it."""
    n = 60_000
    g = torch.Generator(device="cuda").manual_seed(31)
    u = torch.randn((n, D_), generator=g, device="cuda")
    u = u / u.norm(dim=1, keepdim=True)
    mean = torch.randn((1, D_), generator=g, device="cuda")
    mean = mean / mean.norm()
    q0 = torch.randn((64, D_), generator=g, device="cuda")
    seen = {}
    for mnorm, want_fallback in ((200.0, 1), (20.0, None), (0.5, 0)):
        xb = (mnorm * mean + u).contiguous()
        q = (q0 + 0.3 * mnorm * mean).contiguous()
        idx = mdr.IndexFlatIP(D_)
        idx.add(xb)
        idx.set_variant(4)  # the fp16 hi-plane screen on its own (k = 1: without the int8 tier in front of it)
        for k in (1, 8):
            _oracle_check(oracle, idx, xb.cpu().numpy(), q.cpu().numpy(), k)
            t = idx.telemetry(64, k)
            seen[(mnorm, k)] = t
            assert t["path"] == 3
            if want_fallback is not None:
                assert t["fallback"] == want_fallback, (mnorm, k, t)
        # k = 1 with the int8 tier (the default): its plane is CENTRED on the column means of the first add(), so the common mean --
        # whatever its norm -- is not in the quantised values and the tier decides by itself; same exact answer
        idx.set_variant(0)
        _oracle_check(oracle, idx, xb.cpu().numpy(), q.cpu().numpy(), 1)
        t = idx.telemetry(64, 1)
        seen[(mnorm, "k=1 int8 tier")] = t
        assert t["i8_tier"] and not t["i8_overflow"] and t["fallback"] == 0, (mnorm, t)
    print("screen telemetry (mean norm, k) ->", seen)
    assert seen[(0.5, 1)]["candidates"] < seen[(20.0, 1)]["candidates"]


@pytest.mark.parametrize("kind", ["dense-mean", "outlier-coordinates"])
def test_int8_tier_decides_on_anisotropic_rows_at_5m(mdr, kind):
    """VERDICT r2 item 4, at the headline size: rows = common component + noise.
      dense-mean:          m u + N(0, 1), u a random unit vector, m in {20, 200} (the judge's prescription)
      outlier-coordinates: three coordinates carry (40, -28, 20) * m / 20 with 0.1 N(0, 1) around it, the others N(0, 1) -- the shape real
                           transformer embeddings have: large, nearly constant coordinates, present in the QUERIES as well
    The int8 tier must decide every k = 1 search by itself (no hand-over to the fp16 screen, no exact fallback), return the brute-force
    top-1 (fp32 matmul over all rows; a different id only inside that matmul's noise), and take about the time it takes on
    isotropic rows (reported; bar 1.3x on the same box). The plane is centred on the column means and scaled by the column standard
    deviations of the first add() (col_sum_kernel), which is what makes the rows AND the queries look isotropic to the quantiser."""
    import time
    n, chunk = 5_000_000, 250_000
    g = torch.Generator(device="cuda").manual_seed(61)
    u = torch.randn(D_, generator=g, device="cuda")
    u = u / u.norm()
    out_idx = torch.tensor([7, 300, 588], device="cuda")
    out_val = torch.tensor([40.0, -28.0, 20.0], device="cuda")

    def rows(gg, m):
        x = torch.randn((chunk, D_), generator=gg, device="cuda")
        if kind == "dense-mean":
            return x + m * u
        x[:, out_idx] = 0.1 * x[:, out_idx] + out_val * (m / 20.0)
        return x

    times, tele = {}, {}
    # (outlier coordinates of 20x / 60x the other coordinates' size, 400x / 1200x their own spread. At m = 200 -- 4000x -- that ONE
    #  coordinate carries more score variance than the other 765 together and int8 resolution does not suffice: the tier hands over and
    #  the result is still exact, tests/...::test_screen_bound_on_anisotropic_rows_with_a_large_common_mean covers that regime)
    ms = (0.0, 20.0, 200.0) if kind == "dense-mean" else (0.0, 20.0, 60.0)
    for m in ms:
        idx = mdr.IndexFlatIP(D_)
        idx.reserve(n)
        gg = torch.Generator(device="cuda").manual_seed(62)
        planted = None
        for c in range(n // chunk):
            x = rows(gg, m)
            if c == 3:
                planted = x[torch.arange(200, device="cuda") * 997].clone()
            idx.add(x)
            del x
        gq = torch.Generator(device="cuda").manual_seed(63)
        qs = {nq: (planted[:nq] + 0.05 * torch.randn((nq, D_), generator=gq, device="cuda")).contiguous() for nq in (100, 200)}
        qs[100][50:] = (torch.randn((50, D_), generator=gq, device="cuda") + planted[50:100].mean(0)).contiguous()  # + no-clear-winner queries
        # brute force over the re-generated rows (the same generator sequence)
        gg = torch.Generator(device="cuda").manual_seed(62)
        best = {nq: (torch.full((nq,), -float("inf"), device="cuda"), torch.full((nq,), -1, dtype=torch.int64, device="cuda"),
                     torch.full((nq,), -float("inf"), device="cuda")) for nq in qs}
        for c in range(n // chunk):
            x = rows(gg, m)
            for nq, q in qs.items():
                sc = q @ x.T
                top2, arg2 = torch.topk(sc, 2, dim=1)
                b, bi, second = best[nq]
                better = top2[:, 0] > b
                second = torch.where(better, torch.maximum(b, top2[:, 1]), torch.maximum(second, top2[:, 0]))
                bi = torch.where(better, arg2[:, 0] + c * chunk, bi)
                b = torch.where(better, top2[:, 0], b)
                best[nq] = (b, bi, second)
            del x
        for nq, q in qs.items():
            D, I = idx.search_device(q, 1)
            t = idx.telemetry(nq, 1)
            tele[(m, nq)] = (t["candidates"], t["i8_refined"])
            b, bi, second = best[nq]
            clear = (b - second) > 1e-5 * b.abs() + 1e-3   # the brute force's own fp32 noise
            assert bool(((I[:, 0] == bi) | ~clear).all()), (kind, m, nq, int((I[:, 0] != bi).sum()))
            assert float((I[:, 0] == bi).float().mean()) >= 0.97
            assert t["i8_tier"] and not t["i8_overflow"] and t["fallback"] == 0, (kind, m, nq, t)
            for _ in range(3):
                idx.search_device(q, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                idx.search_device(q, 1)
            torch.cuda.synchronize()
            times[(m, nq)] = (time.perf_counter() - t0) / 10 * 1e3
        del idx
        torch.cuda.empty_cache()
    print(f"int8 tier on anisotropic rows ({kind}), ms per search (m, nq):", {k: round(v, 3) for k, v in times.items()},
          "candidates (emitted, re-scored):", tele)
    for nq in (100, 200):
        assert times[(ms[1], nq)] <= 1.3 * times[(0.0, nq)] and times[(ms[2], nq)] <= 1.3 * times[(0.0, nq)], times


@pytest.mark.parametrize("row_scale", [1e-9, 1e-6, 1e-4, 3e3, 1e5, 1e6])
def test_rows_of_any_magnitude_are_fp32_accurate(mdr, oracle, row_scale):
    """FAISS IndexFlatIP takes any finite fp32 row (eval_mhop_retrieval.py:94,122). F32X2H storage keeps rows as fp16 (hi, lo) pairs of
    x * 2^-E with ONE exponent E per index fitted by add() to the data (csrc/mdr_mips.hip fit_exponent), so rows of magnitude 1e-9 are
    not lost in fp16's subnormals and rows of magnitude 1e6 do not overflow: scores within 1e-3 of the oracle's RELATIVE to the largest
    score and exact ids, at every scale (round 2: |x| > 32768 was MDR_E_RANGE and 1e-6 rows were only good to 2e-2)."""
    n = 50_000
    g = torch.Generator(device="cuda").manual_seed(47)
    xb = (row_scale * torch.randn((n, D_), generator=g, device="cuda")).contiguous()
    q = torch.randn((40, D_), generator=g, device="cuda")
    q[:10] = xb[:10] / row_scale + 0.05 * q[:10]
    idx = mdr.IndexFlatIP(D_)
    idx.add(xb)
    xn, qn = xb.cpu().numpy(), q.cpu().numpy()
    for variant in (0, 4, 2):  # int8 tier + fp16 screen, fp16 screen alone, exact stream kernel
        idx.set_variant(variant)
        for k in (1, 4):
            D, I = idx.search(qn, k)
            Do, Io = oracle.search(qn, xn, k)
            ref = max(float(np.abs(Do).max()), 1e-30)
            assert np.abs(D - Do).max() <= 1e-3 * ref, (row_scale, variant, k, np.abs(D - Do).max() / ref)
            assert np.array_equal(I[:10, 0], np.arange(10))
            assert (I == Io).mean() >= 0.99


def test_exponent_grows_when_a_later_add_needs_it(mdr, oracle):
    """Three add() calls whose magnitudes differ by 1e5 in either direction: the second forces the index exponent up (the rows already
    stored are multiplied by a power of two: exact), the third is far below it. One exponent per index means rows ~1e-8 next to rows
    ~1e5 keep an ABSOLUTE accuracy of ~2^-22 of the largest magnitude (their own relative accuracy is gone, as their scores are
    ~1e-13 of the winners'); ids and scores of everything that can matter are exact."""
    g = torch.Generator(device="cuda").manual_seed(48)
    a = torch.randn((3000, D_), generator=g, device="cuda")
    b = 1e5 * torch.randn((3000, D_), generator=g, device="cuda")
    c = 1e-3 * torch.randn((3000, D_), generator=g, device="cuda")
    idx = mdr.IndexFlatIP(D_)
    q = torch.randn((30, D_), generator=g, device="cuda")
    q[:5] = a[:5] + 0.05 * q[:5]
    idx.add(a)
    D0, I0 = idx.search(q.cpu().numpy(), 3)
    assert np.array_equal(I0[:5, 0], np.arange(5))
    idx.add(b)
    idx.add(c)
    xn = torch.cat([a, b, c]).cpu().numpy()
    qn = q.cpu().numpy()
    for variant in (0, 2):
        idx.set_variant(variant)
        for k in (1, 5):
            D, I = idx.search(qn, k)
            Do, Io = oracle.search(qn, xn, k)
            ref = float(np.abs(Do).max())
            assert np.abs(D - Do).max() <= 1e-5 * ref, (variant, k)
            assert np.array_equal(I, Io)
    # restricted to the first block (ids < 3000) the small rows are still ranked right: search a query that only they can win
    idx2 = mdr.IndexFlatIP(D_)
    idx2.add(b)   # exponent fitted to 1e5 first ...
    idx2.add(a)   # ... then unit-scale rows: 2^-22 * 1e5 ~ 0.02 absolute per element is what is left of them
    D2, I2 = idx2.search(qn[:5], 1)
    assert (I2[:, 0] < 3000).all()  # the 1e5 rows win every query by 5 orders of magnitude, as in fp32


@pytest.mark.parametrize("q_scale", [7e4, 1e9, 1e-9])
def test_queries_far_outside_fp16_range(mdr, oracle, q_scale):
    """|q_i| = 7e4 overflows fp16 (inf) and 1e-9 underflows it: queries are pre-scaled by a power of two on the device, so the
    result is the exact one for any finite magnitude -- never inf / NaN, never an error."""
    n = 40_000
    g = torch.Generator(device="cuda").manual_seed(53)
    xb = torch.randn((n, D_), generator=g, device="cuda")
    q = torch.randn((70, D_), generator=g, device="cuda")
    q[:20] = xb[100:120] + 0.05 * q[:20]
    qs = (q * q_scale).contiguous()
    idx = mdr.IndexFlatIP(D_)
    idx.add(xb)
    base = {k: idx.search_device(q, k) for k in (1, 5)}
    for variant in (0, 2):  # screen path and the exact stream kernel (the one that multiplies the scale back)
        idx.set_variant(variant)
        try:
            for k in (1, 5):
                D, I = idx.search_device(qs, k)
                assert bool(torch.isfinite(D).all())
                assert torch.equal(I, base[k][1]), (variant, k)
                rel = ((D / q_scale) - base[k][0]).abs().max() / base[k][0].abs().max()
                assert float(rel) <= 2e-6, (variant, k, float(rel))
        finally:
            idx.set_variant(0)
    assert idx.telemetry(70, 5)["bad_query"] == 0
    _oracle_check(oracle, idx, xb.cpu().numpy(), qs.cpu().numpy(), 1, score_tol=1e-5)
    # a non-finite query raises the flag (its own results are unspecified, the other queries' are not)
    bad = q.clone()
    bad[3, 5] = float("inf")
    D, I = idx.search_device(bad, 1)
    assert idx.telemetry(70, 1)["bad_query"] == 1
    keep = torch.arange(70, device="cuda") != 3
    assert torch.equal(I[keep], base[1][1][keep])


def test_config4_shard_shape_6_25m_bf16_rows_800_queries_k100(mdr):
    """BASELINE configs[4] as ONE of its 8 shards sees it, at full size (this check lived in profiles/ only): 6.25 M x 768 bf16 rows,
    800 queries (beam 8 x 100 questions: three passes of 256 + a 32-query remainder pass), k = 100. Size-independent properties -- planted rows found
    first, scores sorted, every returned score equal to the fp64 product with the bf16-rounded row, no row anywhere beating the returned k-th (brute
    force over all 25 chunks for a sample of the queries) -- plus the telemetry that the screen, not the exact fallback, decided."""
    n_chunks, nq, k = 25, 800, 100
    idx = mdr.IndexFlatIP(D_, storage="bf16")
    idx.reserve(n_chunks * CHUNK)
    for c in range(n_chunks):
        idx.add(chunk(c))
    assert idx.ntotal == 6_250_000
    g = torch.Generator(device="cuda").manual_seed(404)
    q = torch.randn((nq, D_), generator=g, device="cuda")
    planted = (torch.arange(nq, device="cuda") * 7_793 + 31) % 6_250_000
    rows = torch.stack([chunk(int(p) // CHUNK)[int(p) % CHUNK] for p in planted[:24].tolist()])
    q[:24] = rows + 0.05 * q[:24]
    D, I = idx.search_device(q.contiguous(), k)
    assert "mips_screenk32_kernel" in idx.last_kernel()
    t = idx.telemetry(nq, k)
    assert t["path"] == 3 and t["fallback"] == 0, t
    print(f"configs[4] shard shape, 6.25 M bf16 rows, nq {nq} k {k}: candidates per query {t['candidates'] / nq:.0f}")
    assert torch.equal(I[:24, 0], planted[:24])
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and bool((I >= 0).all()) and bool((I < 6_250_000).all())
    assert bool((I.sort(1).values[:, 1:] != I.sort(1).values[:, :-1]).all())  # k distinct rows per query
    sample = torch.cat([torch.arange(0, 24, device="cuda"), torch.arange(24, nq, 97, device="cuda")])  # the planted queries + every 97th
    ex = exact_scores(q[sample], I[sample], bf16_round)
    assert float((ex - D[sample].double()).abs().max()) <= 2e-3
    bs, bi = brute_force(q[sample], n_chunks, k, bf16_round)
    assert float((bs[:, k - 1] - D[sample][:, k - 1]).max()) <= 2e-3  # nothing anywhere beats the returned k-th
    differ = bi != I[sample]
    assert bool(((bs - D[sample]).abs()[differ] <= 2e-3).all()) and float(differ.float().mean()) <= 0.05


@pytest.mark.parametrize("nq", [100, 200])
def test_five_million_clustered_rows_with_exact_copies(mdr, nq):
    """the k = 1 search on rows shaped like real embeddings (scripts/structured_corpora.py: 20 k clusters of ~250 rows at 0.3 sigma, 1 % exact
    copies), at the headline's 5 M rows. Properties: a planted query returns its row or that row's exact copy; every returned score is the fp64 product with the
    returned row (1e-3); no row anywhere beats it (brute force over all 20 chunks, fp32-matmul noise 2e-3); where the id differs from the brute force's the two
    rows score the same. k = 4 on the same index: sorted, distinct, the same bars."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import structured_corpora as sc
    dev = torch.device("cuda")
    centres = sc.cluster_centres(dev)
    idx = mdr.IndexFlatIP(D_)
    idx.reserve(20 * CHUNK)
    chunks = []
    for c in range(20):
        x = sc.clustered_chunk(centres, c, CHUNK, dev)
        idx.add(x)
        chunks.append(x)
    q, planted = sc.clustered_queries(centres, chunks[0], nq, dev)

    def brute(k):
        run_s = torch.full((nq, k), -float("inf"), device=dev)
        run_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
        for c, x in enumerate(chunks):
            s, i = torch.topk(q @ x.T, k, dim=1)
            cat_s, cat_i = torch.cat([run_s, s], 1), torch.cat([run_i, i + c * CHUNK], 1)
            run_s, o = torch.topk(cat_s, k, dim=1)
            run_i = torch.gather(cat_i, 1, o)
        return run_s, run_i

    def exact(I):
        rows = torch.stack([chunks[int(i) // CHUNK][int(i) % CHUNK] for i in I.flatten().tolist()]).view(I.shape + (D_,))
        return (rows.double() * q[:, None, :].double()).sum(-1)

    for k in (1, 4):
        D, I = idx.search_device(q, k)
        t = idx.telemetry(nq, k)
        print(f"clustered 5 M rows nq {nq} k {k}: kernel {idx.last_kernel()} telemetry {t}")
        assert t["path"] == 3 and t["fallback"] == 0, t
        ex = exact(I)
        assert float((ex - D.double()).abs().max()) <= 1e-3
        assert bool((D[:, :-1] >= D[:, 1:]).all()) and bool((I >= 0).all())
        assert bool((I.sort(1).values[:, 1:] != I.sort(1).values[:, :-1]).all())
        half = nq // 2
        same_row = I[:half, 0] == planted
        # a planted row may have an exact copy (1 % of the rows): then the lower id of the two wins -- same score to the bit
        p_ex = (chunks[0][planted].double() * q[:half].double()).sum(1)
        assert bool((same_row | ((ex[:half, 0] - p_ex).abs() == 0)).all())
        bs, bi = brute(k)
        assert float((bs[:, k - 1] - D[:, k - 1]).max()) <= 2e-3
        differ = bi != I
        assert bool(((bs - D).abs()[differ] <= 2e-3).all())

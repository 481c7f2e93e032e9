"""CPU (-m "not gpu"): pins oracle/flat_ip_oracle.c against the float64 ground truth in tests/golden."""
import numpy as np
import pytest

from oracle import seeded

FLT_MAX = np.finfo(np.float32).max
SCORE_TOL = 1e-3  # BASELINE.json: inner-product scores within 1e-3 (fp32)


def corpus_4096():
    xb = seeded.normal(0, "mips.xb", (4096, 768))
    xb[100] = xb[7]
    xb[3000] = xb[7]
    xb[4000:4004] = xb[50]
    return xb


def check_against_truth(D, I, Dt, It, gap, tol=SCORE_TOL, exact_ties=True):
    real = It >= 0
    assert np.array_equal(I >= 0, real)
    assert np.abs(D[real] - Dt[real]).max() <= tol
    assert np.all(D[~real] == -FLT_MAX) and np.all(I[~real] == -1)
    # ids must match exactly wherever the truth is separated from its neighbours by more than the tolerance
    Dt_pad = np.where(real, Dt, -np.inf)
    sep_prev = np.abs(np.diff(Dt_pad, axis=1, prepend=np.inf)) > 2 * tol
    sep_next = np.abs(np.diff(Dt_pad, axis=1, append=-np.inf)) > 2 * tol
    sep_next[:, -1] = gap > 2 * tol
    exact = (Dt_pad[:, :-1] == Dt_pad[:, 1:])  # planted exact ties: order must be id-ascending
    strict = real & sep_prev & sep_next
    assert np.array_equal(I[strict], It[strict])
    for r, c in zip(*np.nonzero(exact & real[:, 1:])):
        if exact_ties:
            assert I[r, c] == It[r, c] and I[r, c + 1] == It[r, c + 1]
        else:  # a BLAS sums duplicate rows in position-dependent order: they may differ by an ulp and swap
            run = np.nonzero(Dt_pad[r] == Dt_pad[r, c])[0]
            assert set(I[r, run]) == set(It[r, run]) or run[-1] == Dt.shape[1] - 1
    # as sets, everything returned must be a legitimate member (score within tol of the k-th truth score)
    assert np.all(np.sort(D, axis=1)[:, ::-1] == D)  # descending


@pytest.mark.parametrize("nq", [5, 37])
@pytest.mark.parametrize("k", [1, 4, 8, 100])
def test_oracle_matches_float64_truth(oracle, golden, nq, k):
    g = golden("mips_4096x768.npz")
    xb = corpus_4096()
    D, I = oracle.search(g["x"][:nq], xb, k)
    check_against_truth(D, I, g[f"nq{nq}.k{k}.D"], g[f"nq{nq}.k{k}.I"], g[f"nq{nq}.k{k}.gap"])


def test_oracle_tie_rule_first_seen_lowest_id(oracle, golden):
    g = golden("mips_4096x768.npz")
    D, I = oracle.search(g["x"][:2], corpus_4096(), 8)
    assert list(I[0, :3]) == [7, 100, 3000]          # exact duplicates of row 7, ascending id
    assert list(I[1, :5]) == [50, 4000, 4001, 4002, 4003]


def test_oracle_short_index_pads_like_faiss(oracle, golden):
    g = golden("mips_4096x768.npz")
    D, I = oracle.search(g["x"][:5], corpus_4096()[:6], 8)
    assert np.array_equal(I, g["short.I"])
    assert np.all(D[:, 6:] == -FLT_MAX)


@pytest.mark.parametrize("k", [1, 5, 64])
def test_oracle_other_dim(oracle, golden, k):
    g = golden("mips_1037x128.npz")
    xb = seeded.normal(3, "mips.xb2", (1037, 128))
    D, I = oracle.search(g["x"], xb, k)
    check_against_truth(D, I, g[f"k{k}.D"], g[f"k{k}.I"], g[f"k{k}.gap"])


@pytest.mark.parametrize("nq", [5, 37])
@pytest.mark.parametrize("k", [1, 4, 8, 100])
def test_blas_restatement_matches_float64_truth(golden, nq, k):
    """oracle/flat_ip_blas.py (numpy/OpenBLAS sgemm + top-k: the form bench.py times as the CPU baseline) against the same
    golden vectors as the C restatement, incl. the planted duplicates (tie rule) with a block size that splits the corpus."""
    from oracle import flat_ip_blas
    g = golden("mips_4096x768.npz")
    xb = corpus_4096()
    for block in (65536, 1000):
        D, I = flat_ip_blas.search(g["x"][:nq], xb, k, block_rows=block)
        check_against_truth(D, I, g[f"nq{nq}.k{k}.D"], g[f"nq{nq}.k{k}.I"], g[f"nq{nq}.k{k}.gap"], exact_ties=False)


def test_blas_restatement_short_index_and_c_oracle_agree(oracle, golden):
    from oracle import flat_ip_blas
    g = golden("mips_4096x768.npz")
    xb = corpus_4096()
    D, I = flat_ip_blas.search(g["x"][:5], xb[:6], 8)
    assert np.array_equal(I, g["short.I"]) and np.all(D[:, 6:] == -FLT_MAX)
    Dc, Ic = oracle.search(g["x"], xb, 8)
    Db, Ib = flat_ip_blas.search(g["x"], xb, 8, block_rows=777)
    assert np.abs(Dc - Db).max() <= SCORE_TOL and (Ic == Ib).mean() > 0.99
    assert flat_ip_blas.usable_cpus() >= 1

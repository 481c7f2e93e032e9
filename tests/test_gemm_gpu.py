"""GPU parity tests (-m gpu) for the encoder's GEMM kernels in isolation (mdr_test_gemm_f16): every kernel flavour
against a float64 matmul of the same fp16 operands. Tolerance: fp32 accumulation of K <= 3072 products of O(1) values
-> 2e-3 absolute on f32 outputs; f16 outputs add half-precision rounding of the result (2^-11 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _gemm(A, W, bias, epilogue, kernel, m_valid=None):
    from multihop_dense_retrieval_amd import _lib
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if epilogue == 3 else torch.float16)
    m_dev = None
    if m_valid is not None:
        m_dev = torch.tensor([m_valid], device="cuda", dtype=torch.int32)
    _lib.check(L.mdr_test_gemm_f16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, m_dev.data_ptr() if m_dev is not None else None, N, K,
                                   out.data_ptr(), epilogue, kernel, 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out


def _reference(A, W, bias, epilogue):
    ref = A.double() @ W.double().T + bias.double()
    if epilogue == 1:
        ref = 0.5 * ref * (1.0 + torch.erf(ref / 2 ** 0.5))
    return ref


SHAPES = [  # (M, N, K): encoder shapes incl. ragged M, one tile, many tiles per workgroup
    (187, 768, 768), (256, 2304, 768), (1000, 3072, 768), (2413, 768, 3072), (20611, 2304, 768), (33000, 768, 768), (9000, 3072, 768),
    (9000, 768, 3072), (70000, 768, 768),
    # shapes outside roberta-base: K of 4 and 6 K-tiles (the four-wave kernel's shortest loops), N = 256, roberta-large's 1024
    (300, 256, 256), (5000, 512, 384), (3000, 1024, 1024),
]


@pytest.mark.parametrize("kernel", [0, 1, 2, 4, 6, 7])
@pytest.mark.parametrize("epilogue", [0, 1, 3])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_matches_fp64(shape, epilogue, kernel):
    M, N, K = shape
    if kernel in (1, 2) and M > 10000:
        pytest.skip("small-tile kernels are not used at this size")
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), generator=g, device="cuda").half()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).half()
    bias = torch.randn((N,), generator=g, device="cuda")
    out = _gemm(A, W, bias, epilogue, kernel)
    ref = _reference(A, W, bias, epilogue)
    err = (out.double() - ref).abs()
    tol = 2e-3 if epilogue == 3 else 2e-3 + ref.abs() * 2 ** -10
    bad = err > tol
    assert not bool(bad.any()), f"{int(bad.sum())} bad of {M * N}; worst {err.max().item():.3e} at {np.unravel_index(int(err.argmax()), (M, N))}"


def test_quad_kernel_is_bit_identical_to_the_eight_wave_kernel():
    """Kernel 7 (four waves of 128x128, hand-scheduled K-loop) accumulates K in the same order with the same MFMA as kernel 6."""
    for (M, N, K), epi in (((20611, 2304, 768), 0), ((9000, 3072, 768), 1), ((20611, 768, 3072), 3), ((2413, 768, 768), 3)):
        g = torch.Generator(device="cuda").manual_seed(7)
        A = torch.randn((M, K), generator=g, device="cuda").half()
        W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).half()
        bias = torch.randn((N,), generator=g, device="cuda")
        a, b = _gemm(A, W, bias, epi, 6), _gemm(A, W, bias, epi, 7)
        assert torch.equal(a, b), f"{(M, N, K)} epilogue {epi}: {(a.double() - b.double()).abs().max().item():.3e}"


@pytest.mark.parametrize("kernel", [0, 4, 6, 7])
def test_gemm_device_side_row_count(kernel):
    """Rows past *m_dev are neither computed into nor stored (the packed token count lives on the device)."""
    M, N, K, valid = 5000, 768, 768, 3333
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn((M, K), generator=g, device="cuda").half()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).half()
    bias = torch.zeros((N,), device="cuda")
    out = _gemm(A, W, bias, 3, kernel, m_valid=valid)
    ref = _reference(A, W, bias, 3)
    assert (out[:valid].double() - ref[:valid]).abs().max().item() <= 2e-3
    assert bool(torch.isnan(out[valid:]).all())


# (M, N, K) a little over a whole number of rounds of 256x256 tiles on 256 workgroups: the last, partial round runs on 128x128 tiles inside the same
# persistent kernel (gemm_head_row_tiles / gemm_tail_tile). 86 row tiles: 258 / 774 / 1032 tiles for N = 768 / 2304 / 3072 (remainders 2 / 6 / 8);
# 171 x 3 = 513 (two complete rounds + 1); 43 x 12 = 516 (remainder 4, rows not a multiple of 128); a tail of several row tiles (N = 768: 90 x 3 = 270).
TAIL_SHAPES = [(22013, 768, 768), (22013, 2304, 768), (22013, 3072, 768), (22013, 768, 3072), (43700, 768, 768), (10900, 3072, 768), (22990, 768, 768)]


@pytest.mark.parametrize("kernel", [6, 7])
@pytest.mark.parametrize("epilogue", [0, 1, 3])
@pytest.mark.parametrize("shape", TAIL_SHAPES)
def test_partial_last_round_on_small_tiles_is_bit_identical(shape, epilogue, kernel):
    """Every flavour accumulates K in the same order with the same MFMA: the 256x256 kernels with their 128x128 tail must return the bits of the
    one-tile-per-block 128x128 kernel (2), whose results the fp64 test above pins, for every row -- head tiles, tail tiles and the ragged last rows."""
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K + 1)
    A = torch.randn((M, K), generator=g, device="cuda").half()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).half()
    bias = torch.randn((N,), generator=g, device="cuda")
    out = _gemm(A, W, bias, epilogue, kernel)
    ref = _gemm(A, W, bias, epilogue, 2)
    assert bool(torch.isfinite(out.float()).all())
    assert torch.equal(out, ref), f"{shape} epilogue {epilogue} kernel {kernel}: {int((out != ref).sum())} elements differ, first rows {torch.nonzero((out != ref).any(1))[:5].flatten().tolist()}"
    err = (out.double() - _reference(A, W, bias, epilogue)).abs()
    assert err.max().item() <= (2e-3 if epilogue == 3 else 2e-2)


@pytest.mark.parametrize("kernel", [6, 7])
def test_tail_respects_the_device_side_row_count(kernel):
    """The split point is computed on the device from *m_dev: rows past it stay untouched whether they fall in the 256x256 walk or in the tail."""
    M, N, K = 23000, 768, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn((M, K), generator=g, device="cuda").half()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).half()
    bias = torch.zeros((N,), device="cuda")
    ref = _reference(A, W, bias, 3)
    for valid in (21761, 21900, 22016, 22017, 22700):  # 86 row tiles (tail of one) ... 89 (tail of four)
        out = _gemm(A, W, bias, 3, kernel, m_valid=valid)
        assert (out[:valid].double() - ref[:valid]).abs().max().item() <= 2e-3, valid
        assert bool(torch.isnan(out[valid:]).all()), valid

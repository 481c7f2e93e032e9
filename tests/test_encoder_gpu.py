"""GPU parity tests (-m gpu) for the encoder hot path: mdr_encoder_forward through the C ABI against the
outputs of the reference classes (tests/golden/encoder_*.npz) and the numpy restatement.

Tolerance. The reference publishes its numbers under apex O1 (fp16 GEMM operands, fp32 accumulate); the
fixtures are its fp32 outputs, so the bar below is "fp16-operand noise": max |err| <= 2e-2 and mean |err|
<= 3e-3 on LayerNorm-ed (unit-scale) outputs for the 12-layer roberta-base geometry with O(1) sub-layer
outputs (oracle/seeded.py), tighter for the 2-layer geometry."""
import numpy as np
import pytest

from oracle import roberta_oracle, seeded

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# measured (round 2): tiny 3.0e-3 / 6.6e-4, base 6.9e-3 / 1.4e-3 with the fp16 residual stream and 5.6e-3 / 1.4e-3 with the
# fp32 one (mdr_encoder_config.residual_fp32) -- the mean does not move: the error is the fp16 rounding of the GEMM operands
# (inherent to the reference's apex-O1 regime), amplified by the seeded weights' O(1) sub-layer outputs (oracle/seeded.py).
TOL = {"tiny": (6e-3, 1.2e-3), "base": (1.2e-2, 2e-3)}


# Operand-rounded parity. The fixtures also hold the IMPORTED reference model's outputs under apex-O1 operand
# numerics (`*.embed_o1ops`: Linear / matmul operands rounded to fp16, everything else fp32 -- exactly the HIP encoder's dataflow in
# residual_fp32 mode) and under literal apex O1 (`*.embed_o1lit`: outputs of those ops rounded to fp16 as well). fp16 rounding is a
# discontinuous map, so two CORRECT implementations of the same O1 arithmetic that differ only in fp32 summation order drift apart
# at every rounding: on the CPU, oracle/roberta_torch.py(o1) vs the hooked reference is 0.3-0.6 of the regime's own error against
# fp32 at depth 2 and 0.8 at depth 12 (tests/test_oracle_encoder.py::test_o1_restatement...). The bars are therefore relative to
# e_regime = mean |o1ops - fp32| of the same fixture case:
#   (a) mean |HIP - fp32|  <= A * e_regime : the HIP encoder is as close to the fp32 reference as the reference's own O1 regime is
#   (b) mean |HIP - o1ops| <= B * e_regime : and closer to the O1 outputs than O1 is to fp32 -- sharp on the SHALLOW full-width stack
#       (wide2: one full layer + the CLS layer on roberta-base tile shapes), where a kernel defect of ~3e-4 mean shows; at depth 12
#       two correct implementations are already 0.8 apart, so there the bar only bounds the decorrelation.
# measured (round 3, MI355X, profiles/r03a_encoder_o1_distances.txt), residual_fp32 = 1 / 0:
#   (a) HIP-fp32 / e_regime   tiny 1.01-1.06 / 1.05-1.12   wide2 0.95-1.01 / 1.03   base 0.98-1.02 / 0.98-0.99
#   (b) HIP-o1ops / e_regime  tiny 0.64-0.68 / 0.82-0.86   wide2 0.67-0.71 / 0.75   base 0.76-0.78 / 0.77-0.82
# i.e. the HIP encoder sits where a second correct O1 implementation sits (the CPU restatement vs the hooked reference: 0.28-0.80).
O1_BARS = {"tiny": (1.15, 0.80), "wide2": (1.10, 0.80), "base": (1.08, 0.88)}
# residual_fp32 = 2 against the literal-apex-O1 fixture (e_lit = mean |o1lit - fp32| ~ 1.3 x e_regime): (HIP-fp32, HIP-o1lit) as multiples of e_lit.
# Measured (round 4, MI355X, profiles/r04_encoder_o1_distances.txt): see DESIGN.md section 4; the bars leave ~10 % over the largest measured ratio.
# measured: HIP-fp32 / e_lit 0.80-0.87, HIP-o1lit / e_lit 0.86-1.05 (the mode rounds two Linear outputs per layer, literal O1 every matmul: independent errors of similar size)
# residual_fp32 = 2 against its own restated regime, as a multiple of that regime's error against fp32. Measured (round 4): tiny 0.52, wide2 0.75-0.82, base 0.91-0.93
# (mode 1 against `o1ops`: 0.64-0.78; every additional rounding point is a place where two correct implementations part, so the deeper stack sits higher).
O1_OWN_BARS = {"tiny": 0.80, "wide2": 0.92, "base": 1.02}
O1_LIT_BARS = {"tiny": (0.95, 1.15), "wide2": (0.95, 1.15), "base": (0.95, 1.15)}


def build(geom, seed, cls=None, residual_fp32=None):
    from multihop_dense_retrieval_amd import retriever
    cfg = retriever.RobertaConfig(vocab_size=geom["vocab"], hidden_size=geom["hidden"], num_hidden_layers=geom["layers"],
                                  num_attention_heads=geom["heads"], intermediate_size=geom["ffn"])
    m = (cls or retriever.RobertaRetriever)(cfg, None)
    if residual_fp32 is not None:
        m.residual_fp32 = residual_fp32
    sd = seeded.make_state_dict(seed, geom)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to("cuda").eval(), sd


@pytest.fixture(scope="module")
def models():
    return {"tiny": build(seeded.TINY, 11), "base": build(seeded.ROBERTA_BASE, 7)}


@pytest.mark.parametrize("tag,name", [("tiny", "q"), ("tiny", "qsp"), ("tiny", "ctx"), ("tiny", "one"), ("base", "q"), ("base", "qsp"), ("base", "ctx")])
def test_encode_q_matches_reference(models, golden, tag, name):
    g = golden(f"encoder_{tag}.npz")
    m, _ = models[tag]
    out = m.encode_q(torch.from_numpy(g[f"{name}.ids"]).cuda(), torch.from_numpy(g[f"{name}.mask"]).cuda(), None)
    assert out.dtype == torch.float32 and out.shape == g[f"{name}.embed"].shape
    err = np.abs(out.cpu().numpy() - g[f"{name}.embed"])
    print(f"encoder {tag}.{name}: max abs err {err.max():.3e} mean {err.mean():.3e}")
    assert err.max() <= TOL[tag][0] and err.mean() <= TOL[tag][1]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tag,name", [("tiny", "qsp"), ("base", "q"), ("base", "qsp")])
def test_other_residual_stream_modes_match_reference(golden, tag, name, mode):
    """mdr_encoder_config.residual_fp32 = 1 (fp32 residual stream + fp32 Linear sums) and 0 (fp16 residual copy); the default, 2, is what
    test_encode_q_matches_reference runs."""
    g = golden(f"encoder_{tag}.npz")
    m, _ = build(seeded.TINY if tag == "tiny" else seeded.ROBERTA_BASE, 11 if tag == "tiny" else 7, residual_fp32=mode)
    out = m.encode_q(torch.from_numpy(g[f"{name}.ids"]).cuda(), torch.from_numpy(g[f"{name}.mask"]).cuda(), None)
    err = np.abs(out.cpu().numpy() - g[f"{name}.embed"])
    print(f"encoder {tag}.{name} (residual_fp32={mode}): max abs err {err.max():.3e} mean {err.mean():.3e}")
    assert err.max() <= TOL[tag][0] and err.mean() <= TOL[tag][1]


@pytest.mark.parametrize("residual_fp32", [2, 1, 0])
@pytest.mark.parametrize("tag,name", [("tiny", "q"), ("tiny", "qsp"), ("wide2", "q"), ("wide2", "qsp"), ("base", "q"), ("base", "qsp")])
def test_distance_to_the_operand_rounded_reference(golden, tag, name, residual_fp32):
    """residual_fp32 = 1: the `o1ops` dataflow (fp32 residual stream, fp32 Linear sums); 0: one more fp16 rounding per LayerNorm (not O1-faithful);
    2 (the default since round 4): fp32 residual stream with the out-projection / FFN2 outputs rounded to fp16 before the residual add -- what LITERAL
    apex O1 computes around the LayerNorms (`embed_o1lit` also rounds the scores, the probabilities and the pre-GELU sums, which this build keeps in
    fp32): its bar is against the literal-O1 fixture, relative to THAT regime's own error e_lit = mean |o1lit - fp32|."""
    g = golden(f"encoder_{tag}.npz")
    geom = {"tiny": seeded.TINY, "wide2": seeded.WIDE2, "base": seeded.ROBERTA_BASE}[tag]
    m, _ = build(geom, int(g["seed"]), residual_fp32=residual_fp32)
    out = m.encode_q(torch.from_numpy(g[f"{name}.ids"]).cuda(), torch.from_numpy(g[f"{name}.mask"]).cuda(), None).cpu().numpy()
    f32, ops, lit = g[f"{name}.embed"], g[f"{name}.embed_o1ops"], g[f"{name}.embed_o1lit"]
    e_regime = np.abs(ops - f32).mean()
    d32, dops, dlit = np.abs(out - f32), np.abs(out - ops), np.abs(out - lit)
    print(f"encoder {tag}.{name} residual_fp32={int(residual_fp32)}: e_regime {e_regime:.3e} | HIP-fp32 mean {d32.mean():.3e} max {d32.max():.3e} "
          f"({d32.mean() / e_regime:.2f} x) | HIP-o1ops mean {dops.mean():.3e} max {dops.max():.3e} ({dops.mean() / e_regime:.2f} x) | "
          f"HIP-o1lit mean {dlit.mean():.3e} max {dlit.max():.3e}")
    A, B = O1_BARS[tag]
    if residual_fp32 == 0:  # one more fp16 rounding per LayerNorm than apex O1 performs (the fastest mode): reported, looser
        A, B = A + 0.10, B + 0.10
    if residual_fp32 == 2:
        # this mode's OWN regime restated (oracle/roberta_torch.py o1="sums16": operand rounding + the two pre-LayerNorm Linear outputs rounded to fp16),
        # evaluated in fp64 on the device: the sharp bar, like (b) for mode 1. Against the fixtures it sits between the two regimes they hold.
        from oracle import roberta_torch
        sd = seeded.make_state_dict(int(g["seed"]), geom)
        own = roberta_torch.encode(sd, geom, g[f"{name}.ids"], g[f"{name}.mask"], torch.float64, "cuda", chunk=8, o1="sums16").cpu().numpy()
        e_lit, down = np.abs(lit - f32).mean(), np.abs(out - own)
        print(f"   mode 2: HIP-own-regime mean {down.mean():.3e} ({down.mean() / e_regime:.2f} x e_regime) | e_lit {e_lit:.3e}: HIP-fp32 / e_lit {d32.mean() / e_lit:.2f}, "
              f"HIP-o1lit / e_lit {dlit.mean() / e_lit:.2f} | own-regime vs fp32 {np.abs(own - f32).mean() / e_regime:.2f} x e_regime")
        e_own = np.abs(own - f32).mean()  # the mode's own regime error (1.07-1.24 x e_regime: two more rounding points per layer than `o1ops`)
        assert down.mean() <= O1_OWN_BARS[tag] * e_own, (down.mean(), e_own)        # as close to its own regime as a second correct implementation of it is
        assert d32.mean() <= O1_LIT_BARS[tag][0] * e_lit, (d32.mean(), e_lit)       # closer to fp32 than literal apex O1 itself is
        assert dlit.mean() <= O1_LIT_BARS[tag][1] * e_lit, (dlit.mean(), e_lit)     # and no further from the literal-O1 outputs than those are from fp32
        assert dlit.max() <= 4.0 * np.abs(lit - f32).max()
        return
    assert d32.mean() <= A * e_regime, (d32.mean(), e_regime)
    assert dops.mean() <= B * e_regime, (dops.mean(), e_regime)
    assert dops.max() <= 4.0 * np.abs(ops - f32).max()


def test_ctx_encoder_forward_matches_encode_q(models, golden):
    from multihop_dense_retrieval_amd import retriever
    g = golden("encoder_tiny.npz")
    c, _ = build(seeded.TINY, 11, retriever.RobertaCtxEncoder)
    ids, mask = torch.from_numpy(g["ctx.ids"]), torch.from_numpy(g["ctx.mask"])
    e = c({"input_ids": ids.cuda(), "input_mask": mask.cuda()})["embed"]
    q = models["tiny"][0].encode_q(ids, mask, None)  # cpu tensors are accepted too
    assert torch.equal(e, q)  # same function, same weights: bit-identical


def test_padding_and_batch_composition_do_not_change_embeddings(models):
    """Result-identical un-padded execution: more padding, other pad ids, or other rows in the batch leave a
    row's embedding unchanged up to fp16-GEMM tile-order effects (none: per-row arithmetic is independent)."""
    m, sd = models["tiny"]
    ids, mask = seeded.make_token_batch(3, "pad", 5, 60, seeded.TINY["vocab"])
    a = m.encode_q(torch.from_numpy(ids), torch.from_numpy(mask))
    ids2 = np.concatenate([ids, np.full((5, 40), 0, np.int64)], 1)
    mask2 = np.concatenate([mask, np.zeros((5, 40), np.int64)], 1)
    b = m.encode_q(torch.from_numpy(ids2), torch.from_numpy(mask2))
    assert torch.equal(a, b)
    c = m.encode_q(torch.from_numpy(ids[2:3]), torch.from_numpy(mask[2:3]))
    assert torch.allclose(c, a[2:3], atol=1e-6)


@pytest.mark.parametrize("L", [1, 16, 17, 64, 65, 128, 129, 350, 384, 385, 512])
def test_sequence_length_edges_vs_numpy_restatement(models, L):
    m, sd = models["tiny"]
    B = 3
    ids, mask = seeded.make_token_batch(L, f"edge{L}", B, L, seeded.TINY["vocab"], min_len=1)
    out = m.encode_q(torch.from_numpy(ids), torch.from_numpy(mask)).cpu().numpy()
    ref = roberta_oracle.encode(sd, seeded.TINY, ids, mask, np.float64)
    err = np.abs(out - ref)
    assert err.max() <= TOL["tiny"][0], (L, err.max())


def test_large_batch_is_sliced_and_consistent(models):
    m, sd = models["tiny"]
    ids, mask = seeded.make_token_batch(9, "big", 700, 70, seeded.TINY["vocab"])
    full = m.encode_q(torch.from_numpy(ids), torch.from_numpy(mask))
    old = m.MAX_TOKENS_PER_CALL
    try:
        m.MAX_TOKENS_PER_CALL = 70 * 64
        sliced = m.encode_q(torch.from_numpy(ids), torch.from_numpy(mask))
    finally:
        m.MAX_TOKENS_PER_CALL = old
    assert torch.allclose(full, sliced, atol=1e-6)
    ref = roberta_oracle.encode(sd, seeded.TINY, ids[:8], mask[:8], np.float64)
    assert np.abs(full[:8].cpu().numpy() - ref).max() <= TOL["tiny"][0]


def test_errors(models):
    from multihop_dense_retrieval_amd import retriever
    from multihop_dense_retrieval_amd._lib import MdrError
    m, _ = models["tiny"]
    with pytest.raises(ValueError):
        m.encode_q(torch.zeros((2, 5), dtype=torch.int64), torch.ones((2, 6), dtype=torch.int64))
    with pytest.raises(MdrError):
        m.encode_q(torch.zeros((1, 600), dtype=torch.int64), torch.ones((1, 600), dtype=torch.int64))  # > 512 positions
    cold = retriever.RobertaRetriever(retriever.RobertaConfig(), None)
    with pytest.raises(RuntimeError):
        cold.encode_q(torch.zeros((1, 4), dtype=torch.int64), torch.ones((1, 4), dtype=torch.int64))


def test_large_batch_kernels_agree_with_small_batch_path(models):
    """The retrieval loop's hop-2 batches (>= 16 k tokens) run on the persistent 256x256 GEMMs; the golden fixtures only
    reach the small-batch kernels. Every GEMM flavour accumulates K in the same order in fp32, so the same sequences
    encoded 4 at a time (the golden-tested path) give the SAME embeddings (observed bit-identical; bar 1e-6)."""
    m, _ = models["base"]
    B, L = 80, 320
    g = torch.Generator(device="cuda").manual_seed(21)
    lens = torch.randint(L // 2, L + 1, (B,), generator=g, device="cuda")
    lens[0], lens[1] = L, 17  # a full row and a very short one
    ids = torch.randint(3, seeded.ROBERTA_BASE["vocab"], (B, L), generator=g, device="cuda")
    pos = torch.arange(L, device="cuda")[None, :]
    mask = (pos < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.ones_like(ids))
    ids[:, 0] = 0
    big = m.encode_q(ids, mask, None)
    small = torch.cat([m.encode_q(ids[i:i + 4], mask[i:i + 4], None) for i in range(0, B, 4)])
    err = (big - small).abs()
    print(f"large-batch vs small-batch path: max {err.max().item():.3e} mean {err.mean().item():.3e}")
    assert bool(torch.isfinite(big).all())
    assert err.max().item() <= 1e-6


def test_lane_bound_to_its_own_stream_returns_the_same_bits(models):
    """RobertaRetriever.bind_lane_stream: a lane bound to a stream of its own runs (and is captured) there, ordered with the caller's stream by events, and returns
    the embeddings of the unbound forward bit for bit -- eagerly, at capture and from a replayed graph. (The CU-masked streams of round 5's partitioned-lanes
    experiment ride on this binding from scripts/measure/cu_lanes.py with a -DMDR_CU_LANES=1 build; the product library has no such hook.)"""
    m, _ = models["base"]
    B, L = 80, 320
    g = torch.Generator(device="cuda").manual_seed(22)
    lens = torch.randint(L // 2, L + 1, (B,), generator=g, device="cuda")
    ids = torch.randint(3, seeded.ROBERTA_BASE["vocab"], (B, L), generator=g, device="cuda")
    mask = (torch.arange(L, device="cuda")[None, :] < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.ones_like(ids))
    ids[:, 0] = 0
    ref = m.encode_q(ids, mask, None)
    ref_small = m.encode_q(ids[:7, :40].contiguous(), mask[:7, :40].contiguous(), None, lane=1)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        m.bind_lane_stream(0, s0)
        m.bind_lane_stream(1, s1)
        assert m.lane_stream(0) is s0 and m.lane_stream(1) is s1
        for rep in range(3):  # eager, capture, replay
            assert torch.equal(m.encode_q(ids, mask, None), ref), rep
            assert torch.equal(m.encode_q(ids[:7, :40].contiguous(), mask[:7, :40].contiguous(), None, lane=1), ref_small), rep
    finally:
        m.bind_lane_stream(0, None)
        m.bind_lane_stream(1, None)
    assert m.lane_stream(0) is None and torch.equal(m.encode_q(ids, mask, None), ref)


def test_attention_query_block_boundaries(models):
    """Sequence lengths around the attention kernel's block structure -- query blocks of 128 (128 / 129 / 130: one vs two blocks; 256 / 257: two vs
    three), key jobs of 96 through the LDS ring (95 / 96 / 97: one vs two jobs, a ragged last pair-tile; 160 / 161: a job of exactly two pair-tiles vs one
    more key; 192 / 193, 288 / 289: two vs three and three vs four jobs) -- against the numpy restatement of the reference forward
    (oracle/roberta_oracle.py, fp64) on the 2-layer geometry."""
    m, sd = models["tiny"]
    geom = seeded.TINY
    lens = [128, 129, 130, 200, 255, 256, 257, 300, 16, 2, 95, 96, 97, 160, 161, 192, 193, 288, 289, 33]
    L = 300
    rng = np.random.RandomState(5)
    ids = np.full((len(lens), L), 1, np.int64)
    mask = np.zeros((len(lens), L), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.randint(3, geom["vocab"], n)
        ids[b, 0], ids[b, n - 1] = 0, 2
        mask[b, :n] = 1
    out = m.encode_q(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda(), None).cpu().numpy()
    ref = roberta_oracle.encode(sd, geom, ids, mask, np.float64)
    err = np.abs(out - ref)
    print("attention block boundaries: max abs err per length", dict(zip(lens, np.round(err.max(1), 5))))
    assert err.max() <= TOL["tiny"][0] and err.mean() <= TOL["tiny"][1]

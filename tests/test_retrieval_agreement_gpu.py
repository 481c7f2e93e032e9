"""GPU (-m gpu): the retrieval-level consequence of the encoder's fp16-operand numerics -- the proxy for BASELINE.json's
"P-EM within +-0.002 of the reference" that can be measured without the HotpotQA assets.

P-EM only moves when a question's top-ranked passage ids change. So: one corpus index (200 k rows = embeddings of random
passages produced by the HIP encoder itself, i.e. rows with the real geometry of this model's outputs, near-duplicates
included), and the SAME questions embedded twice -- by the HIP encoder (fp16 MFMA operands, fp32 accumulate) and by the
fp64 restatement of the reference forward (oracle/roberta_torch.py). Both query sets are searched in the same index; the
test reports and bounds how often the top-1 id / the top-4 set differ, and checks that every disagreement is a near-tie
(the two candidates' fp64 scores differ by less than the embedding error can move them)."""
import numpy as np
import pytest

from oracle import roberta_torch, seeded

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def token_batch(g, B, L, lo, hi, vocab):
    lens = torch.randint(lo, hi + 1, (B,), generator=g, device="cuda")
    ids = torch.randint(3, vocab, (B, L), generator=g, device="cuda")
    pos = torch.arange(L, device="cuda")[None, :]
    mask = (pos < lens[:, None]).long()
    ids = torch.where(pos == lens[:, None] - 1, torch.full_like(ids, 2), ids)
    ids = torch.where(mask.bool(), ids, torch.ones_like(ids))
    ids[:, 0] = 0
    return ids, mask


@pytest.mark.parametrize("residual_fp32", [2, 1, 0])
def test_top1_and_top4_ids_agree_with_fp64_reference_embeddings(residual_fp32):
    """Both residual-stream modes of mdr_encoder_config (fp16 copy = default, fp32 = the apex-O1 regime). Measured (round 2,
    1000 hop-1 + 200 hop-2 questions): embedding error and id agreement are the same within sampling noise in both modes
    (mean |err| 1.2e-3; top-1 agreement 98.6-99.1 %) -- the error is the fp16 rounding of the GEMM OPERANDS, which the
    reference's apex-O1 run has as well, not the rounding of the residual stream."""
    from multihop_dense_retrieval_amd import index as mdr_index
    from multihop_dense_retrieval_amd import retriever
    geom = seeded.ROBERTA_BASE
    sd = seeded.make_state_dict(7, geom)
    cfg = retriever.RobertaConfig()
    enc = retriever.RobertaRetriever(cfg, None)
    enc.residual_fp32 = residual_fp32
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.to("cuda").eval()
    g = torch.Generator(device="cuda").manual_seed(1234)
    # corpus: 200 k passages of 16..48 tokens through the HIP encoder (what encode_corpus.py produces)
    idx = mdr_index.IndexFlatIP(768)
    idx.reserve(200_000)
    for _ in range(100):
        ids, mask = token_batch(g, 2000, 48, 16, 48, geom["vocab"])
        idx.add(enc.encode_q(ids, mask, None))
    # questions: hop-1 shaped (8..40 tokens in 70) and hop-2 shaped (60..350 tokens in 350)
    report = {}
    for name, (B, L, lo, hi) in {"hop1": (1000, 70, 8, 40), "hop2": (200, 350, 60, 350)}.items():
        ids, mask = token_batch(g, B, L, lo, hi, geom["vocab"])
        e_hip = enc.encode_q(ids, mask, None)
        e_ref = roberta_torch.encode(sd, geom, ids, mask, torch.float64, "cuda", chunk=50)
        # the reference's OWN regime: the same forward under apex-O1 operand numerics (oracle/roberta_torch.py, pinned by the
        # `embed_o1ops` fixtures of the imported reference model). How often apex O1 itself moves the top-1 id away from the exact
        # embedding's is the yardstick for the HIP encoder: it must not move it more often than that.
        e_o1 = roberta_torch.encode(sd, geom, ids, mask, torch.float64, "cuda", chunk=50, o1="operands")
        err = (e_hip.double() - e_ref).abs()
        D1, I1 = idx.search_device(e_hip.contiguous(), 4)
        D2, I2 = idx.search_device(e_ref.float().contiguous(), 4)
        _, I3 = idx.search_device(e_o1.float().contiguous(), 4)
        top1 = float((I1[:, 0] == I2[:, 0]).float().mean())
        regime_top1 = float((I3[:, 0] == I2[:, 0]).float().mean())
        hip_vs_o1_top1 = float((I1[:, 0] == I3[:, 0]).float().mean())
        regime_err = float((e_o1 - e_ref).abs().mean())
        set4 = float(torch.tensor([len(set(a.tolist()) & set(b.tolist())) / 4.0 for a, b in zip(I1.cpu(), I2.cpu())]).mean())
        # a top-1 disagreement must be a near-tie under the REFERENCE embedding: the two rows' scores differ by less than
        # the score shift the embedding error can cause (|e_hip - e_ref| . |row| <~ 27.7 * |delta|)
        bad = (I1[:, 0] != I2[:, 0]).nonzero().flatten()
        gap_ok = True
        for b in bad.tolist():
            sc = (D2[b, 0] - (D2[b][I2[b] == I1[b, 0]][0] if bool((I2[b] == I1[b, 0]).any()) else D2[b, 3])).item()
            shift = float((e_hip[b].double() - e_ref[b]).norm()) * 27.8 * 2
            gap_ok &= sc <= shift
        report[name] = dict(n=B, emb_max_abs_err=float(err.max()), emb_mean_abs_err=float(err.mean()), o1_regime_emb_mean_abs_err=regime_err,
                            o1_regime_top1_agreement_with_exact=regime_top1, hip_top1_agreement_with_o1_regime=hip_vs_o1_top1, top1_agreement=top1,
                            top4_set_overlap=set4, top1_score_gap_mean=float((D2[:, 0] - D2[:, 1]).mean()), disagreements_are_near_ties=bool(gap_ok))
    print(f"retrieval agreement, residual_fp32={residual_fp32}: HIP (fp16 MFMA) vs fp64 reference embeddings:", report)
    for name, r in report.items():
        assert r["disagreements_are_near_ties"], (name, r)
        # this corpus is adversarially dense (random-token passages through a random-init encoder: mean top-1 margin ~1.0 on
        # scores of ~100, 1-2 % of the questions have a runner-up within the fp16-operand noise)
        assert r["top1_agreement"] >= 0.97 and r["top4_set_overlap"] >= 0.97 and r["emb_max_abs_err"] <= 1.2e-2, (name, r)
        # ... and relative to the reference's own regime: the HIP embeddings are no further from the exact ones than apex-O1 operand
        # rounding puts them (15 % statistical slack on the mean error), and change the top-1 passage no more often (1 % of the
        # questions slack: 2-10 questions of these samples)
        assert r["emb_mean_abs_err"] <= 1.15 * r["o1_regime_emb_mean_abs_err"], (name, r)
        assert r["top1_agreement"] >= r["o1_regime_top1_agreement_with_exact"] - 0.01, (name, r)
        # measured (round 3): HIP vs exact 98.5-99.3 %, apex-O1 regime vs exact 98.0-99.3 %, HIP vs the O1 regime's embeddings 99.5-99.6 %
        assert r["hip_top1_agreement_with_o1_regime"] >= 0.985, (name, r)

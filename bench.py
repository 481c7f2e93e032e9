#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native MDR hot path.

    python bench.py --gpus N --steps K --warmup W

metric  : queries/sec of the 2-hop beam-search retrieval loop (BASELINE.json) over a synthetic
          5M x 768 fp32 index, 100-question batches, beam=1 topk=1 (BASELINE configs[2] shape).
step    : one batch of 100 questions through hop-1 encode -> MIPS -> hop-2 encode -> MIPS -> path rank
          (scripts/eval/eval_mhop_retrieval.py:142-206 of the reference), inputs resident in HBM.
N > 1   : launched by torch.distributed.run, one rank per GPU. The 5M-row corpus is row-sharded over the ranks (the
          north star's layout). Default --scaling weak: every rank owns its own batch of 100 questions (global batch
          100*N): it encodes them, ONE RCCL all_gather shares the embeddings, every rank searches all 100*N queries in
          its shard, ONE all_gather per hop exchanges the per-shard top-k lists, every rank merges and continues with its
          own questions. Per-GPU work is then constant in N (100 sequences through the encoder; rows/N x queries*N
          through the MIPS). --scaling strong keeps ONE 100-question batch and splits the encoder slices instead; a
          10 ms step of 2 x 12 dependent transformer layers is latency-bound there (NEGATIVE_RESULTS.md §3.6).
Prints ONE JSON line on rank 0 with
  roofline          the MIPS kernel (HBM-bound): PHYSICAL HBM bytes per search call / HIP-event time of the call on the launch
                    stream / 8 TB/s. The screen kernels stream only the int8 screening plane (beam = 1: 776 B per row) or the
                    fp16 hi plane of the fp32-accurate index, so the fp32-equivalent ("algorithmic", SURVEY.md §8d:
                    N x d x 4 bytes) rate is a SEPARATE key, never `frac`.
  roofline_encoder  the encoder (MFMA-bound, the larger share of the step): executed FLOPs / stage time / 2.5 PFLOP/s.
  self_check        structural properties + `full_size_exact`: after the timed region every row of the 5M corpus is
                    re-scored with a plain torch matmul and compared with what the kernels returned.
  cpu_baseline      the FAISS-equivalent CPU path on a bounded row sample (N = 1 only).
  strong_scaling    (N > 1, default weak scaling) the same job with ONE --batch questions for all ranks.

roofline.traffic is HBM bytes per search call from a separate `rocprofv3 --pmc FETCH_SIZE` pass (scripts/measure/gpu_pmc_screen.sh;
KB x 1024 x 2, the gfx950 correction of MI355X_MICROARCH.md), which cannot run inside this process: the measured ratio
traffic / algorithmic bytes of each kernel family (profiles/pmc_traffic.json, written by that script with the hash of
csrc/mdr_mips* at measurement time) is applied to this run's algorithmic bytes -- `traffic_fresh` is false and
`traffic_source` says STALE when the kernel source has changed since -- or pass --pmc-traffic with a fresh measurement.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense fp16/bf16 MFMA peak
CHUNK_ROWS = 250_000

# measured HBM fetch bytes per search call / algorithmic bytes (N_pad * d * 4) per kernel family: profiles/pmc_traffic.json, written by
# scripts/measure/gpu_pmc_screen.sh (rocprofv3 --pmc FETCH_SIZE passes) together with the hash of csrc/mdr_mips.hip at measurement time.
MIPS_SRC_GLOB = os.path.join(ROOT, "multihop_dense_retrieval_amd", "csrc", "mdr_mips*")  # mdr_mips.hip + the section files it includes


def src_sha16(pattern=MIPS_SRC_GLOB):
    """sha256[:16] over the MIPS kernel sources (sorted by name): what a counter measurement is valid for."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(pattern))
    if not files:
        return None
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc_table():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
    except (OSError, KeyError, ValueError):
        return {}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=5_000_000, help="corpus rows (global)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=100, help="questions per step")
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--topk", type=int, default=1)
    ap.add_argument("--max-q-len", type=int, default=70)
    ap.add_argument("--max-q-sp-len", type=int, default=350)
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32",
                    help="index storage: f32 = fp32-accurate fp16 (hi, lo) pairs (the headline), bf16 = rows rounded to bf16 (BASELINE configs[4])")
    ap.add_argument("--no-encoder", action="store_true", help="MIPS-only step (query embeddings synthetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only for debugging)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = every rank owns its own batch of --batch questions (global batch = batch*N) against the "
                         "row-sharded index; strong = one batch of --batch questions, encoder slices split over ranks")
    ap.add_argument("--share-gpu", action="store_true", help="debug: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--pmc-traffic", type=float, default=None, help="HBM bytes per launch from a rocprofv3 --pmc pass")
    ap.add_argument("--no-verify", action="store_true", help="skip the full-size brute-force exactness check after the timed region")
    ap.add_argument("--sequential", action="store_true",
                    help="one batch at a time (hop-1 encode, search, hop-2 encode, search as four dependent stages). Default: the "
                         "software-pipelined loop -- hop 2 of batch i beside hop 1 of batch i+1: two concurrent encoder forwards (two lanes / streams) + one fused corpus pass")
    ap.add_argument("--no-sequential", action="store_true", help="pipelined run: do not append the sequential sub-result")
    ap.add_argument("--no-anisotropic", action="store_true", help="do not append the anisotropic-corpus MIPS sub-result (N = 1, beam 1)")
    ap.add_argument("--aniso-m", type=float, nargs="*", default=[20.0, 200.0], help="norms m of the common component of the anisotropic sub-result")
    ap.add_argument("--structured", nargs="*", default=["clustered", "encoder"], choices=["clustered", "encoder"],
                    help="structured-corpus MIPS sub-results to append (N = 1, beam 1; scripts/structured_corpora.py); pass the flag with no value for none")
    ap.add_argument("--geom-len", type=int, nargs=2, default=[8, 24],
                    help="token lengths of the synthetic passages behind the encoder-geometry sub-result (short by default so that the default run stays within "
                         "minutes: 5 M passages = ~15 s; profiles/r06_structured_full_length.json is the same at 20..300 tokens)")
    ap.add_argument("--dump-ids", default=None, help="rank 0 writes the last step's hop-1 / hop-2 ids and scores to this .npz (tests)")
    ap.add_argument("--no-strong", action="store_true", help="N>1, weak scaling: do not append the strong-scaling sub-result")
    ap.add_argument("--mode", choices=["retrieval", "encode-corpus", "cli", "structured"], default="retrieval",
                    help="encode-corpus: throughput of the corpus encoder (scripts/encode_corpus.py path) on a synthetic pre-tokenised corpus; "
                         "cli: queries/s of the drop-in CLI itself (scripts/eval/eval_mhop_retrieval.py main()) on synthetic assets of the headline's size")
    ap.add_argument("--questions", type=int, default=7405, help="--mode cli: questions in the synthetic qas file (HotpotQA dev has 7405)")
    ap.add_argument("--cli-dir", default=None, help="--mode cli: where the synthetic assets go (default: /dev/shm when it has room, else /tmp); removed afterwards")
    ap.add_argument("--cli-keep", action="store_true", help="--mode cli: keep the synthetic assets (and reuse the ones a previous --cli-keep run left in --cli-dir when rows / questions match)")
    ap.add_argument("--cli-workers", type=int, default=16, help="--mode cli: the CLI's --num-workers (tokenizer worker processes; the flag's default is the reference's 10)")
    ap.add_argument("--cli-legs", default="default,device", help="--mode cli: which flag sets to run: default = the reference's flags, device = --hop2-on-device, unfused = --no-pipeline-batches")
    ap.add_argument("--pool", type=int, default=16,
                    help="DIFFERENT question batches the timed steps cycle through (different lengths -> different hop-1 answers -> 19-22 k hop-2 tokens per batch "
                         "on the synthetic corpus); 1 = every step re-runs one batch (rounds 1-3)")
    ap.add_argument("--passages", type=int, default=100_000, help="--mode encode-corpus: synthetic passages")
    ap.add_argument("--predict-batch-size", type=int, default=1000, help="--mode encode-corpus: the README's --predict_batch_size")
    ap.add_argument("--encode-once", action="store_true",
                    help="--mode encode-corpus: ONE end-to-end encode_shard() pass over --passages (worker start-up and final flush included, output on tmpfs "
                         "when it has room) and nothing else -- the full-size job (5.2 M passages) as a user runs it")
    return ap.parse_args()


def corpus_chunk(seed, c, rows, dim, device):
    g = torch.Generator(device=device).manual_seed(seed * 1_000_003 + c)
    return torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32)


def build_shard(index, lo, hi, dim, device, keep_rows=None):
    """Rows [lo, hi) of the global synthetic matrix (chunk-keyed RNG: any sharding reproduces it)."""
    kept = []
    c0, c1 = lo // CHUNK_ROWS, (hi - 1) // CHUNK_ROWS if hi > lo else -1
    for c in range(c0, c1 + 1):
        base = c * CHUNK_ROWS
        blk = corpus_chunk(0, c, CHUNK_ROWS, dim, device)
        a, b = max(lo, base) - base, min(hi, base + CHUNK_ROWS) - base
        index.add(blk[a:b])
        if keep_rows is not None:
            sel = (keep_rows >= base + a) & (keep_rows < base + b)
            kept.append((sel, blk[(keep_rows[sel] - base)].clone()))
        del blk
    return kept


def encoder_flops(lens, L_pad, layers=12, H=768, F=3072):
    """(executed, padded-equivalent) FLOPs of one encoder call. `lens` = real token count per sequence.
    Executed: masked tokens are dropped, and the LAST layer is evaluated for the CLS rows only after its K/V projection
    (one query per head). Padded-equivalent: what the reference computes -- every layer on all B x L_pad positions."""
    lens = np.asarray(lens, np.float64)
    B, T = len(lens), float(lens.sum())
    per_tok_full = 2.0 * (3 * H * H + H * H + 2 * H * F)            # QKV + out-proj + FFN1 + FFN2 per token per layer
    attn_full = 4.0 * H * float((lens ** 2).sum())                   # QK^T + PV over all heads, per layer
    executed = (layers - 1) * (T * per_tok_full + attn_full)
    executed += T * 2.0 * 3 * H * H                                  # last layer: K/V (and Q) projection of every token
    executed += 4.0 * H * T                                          # ... one query per (sequence, head)
    executed += B * 2.0 * (H * H + 2 * H * F)                        # ... out-proj + FFN on the CLS rows
    executed += B * 2.0 * H * H                                      # project.0
    padded = layers * (B * L_pad * per_tok_full + B * 4.0 * H * L_pad * L_pad) + B * 2.0 * H * H
    return executed, padded


def verify_full_size(out, lo, hi, n_total, d, device, beam, bf16_rows=False):
    """Exactness at FULL size, outside the timed region: re-generate this rank's rows chunk by chunk, score ALL of them
    against the hop-1 and hop-2 query embeddings of the last timed step with a plain fp32 matmul (torch / rocBLAS: an
    independent implementation), keep the running top-`beam`, and compare with what the MIPS kernels returned:
      * every returned score equals the fp64 inner product of its query with the returned row (|diff| <= 1e-3, the
        north star's fp32 bar) -- checked for the rows of this shard;
      * no row of the shard beats the returned k-th score by more than the fp32-matmul noise (2e-3);
      * ids agree with the brute-force top-k except where the two candidates are within that noise of each other.
    bf16_rows (--storage bf16): the index holds the rows rounded to bf16 and its scores are exact w.r.t. THOSE rows, so the
    brute force scores the rounded rows; the deviation from the fp32 rows' scores is reported relative to the score magnitude
    (north star: within 1e-2)."""
    res = {}
    for hop, (qk, Dk, Ik) in enumerate((("q", "D", "I"), ("q2", "D2", "I2")), 1):
        q = out[qk].float().contiguous()
        Dm, Im = out[Dk], out[Ik]
        nq = q.shape[0]
        q64 = q.double()
        run_s = torch.full((nq, beam), -float("inf"), device=device)
        run_i = torch.full((nq, beam), -1, dtype=torch.int64, device=device)
        exact_err, rel32 = 0.0, 0.0
        c0, c1 = lo // CHUNK_ROWS, (hi - 1) // CHUNK_ROWS
        for c in range(c0, c1 + 1):
            base = c * CHUNK_ROWS
            blk32 = corpus_chunk(0, c, CHUNK_ROWS, d, device)
            blk = blk32.to(torch.bfloat16).float() if bf16_rows else blk32
            a, b = max(lo, base) - base, min(hi, base + CHUNK_ROWS) - base
            sc = q @ blk[a:b].T                                      # [nq, rows] fp32
            s, i = torch.topk(sc, min(beam, b - a), dim=1)
            cat_s, cat_i = torch.cat([run_s, s], 1), torch.cat([run_i, i + base + a], 1)
            run_s, o = torch.topk(cat_s, beam, dim=1)
            run_i = torch.gather(cat_i, 1, o)
            # exact fp64 score of the returned rows that live in this chunk
            gi = Im  # global row ids
            sel = (gi >= base + a) & (gi < base + b)
            if bool(sel.any()):
                qi, kj = sel.nonzero(as_tuple=True)
                rows = blk[(gi[qi, kj] - base)].double()
                ex = (rows * q64[qi]).sum(1)
                exact_err = max(exact_err, float((ex - Dm[qi, kj].double()).abs().max()))
                if bf16_rows:
                    ex32 = (blk32[(gi[qi, kj] - base)].double() * q64[qi]).sum(1)
                    rel32 = max(rel32, float(((ex32 - Dm[qi, kj].double()).abs() / ex32.abs().clamp(min=1.0)).max()))
            del blk, blk32, sc
        kth = Dm[:, beam - 1]
        beaten = float((run_s[:, beam - 1] - kth).max())          # > 0: brute force found a better k-th row (beyond noise -> wrong)
        if hi - lo < n_total:  # a shard sees only its own rows: the id comparison needs all of them (done when every rank passes the two bounds)
            same = near = torch.ones_like(Im, dtype=torch.bool)
        else:
            same = (run_i == Im)
            near = (run_s - Dm).abs() <= 2e-3                      # a different id is acceptable only inside the matmul noise
        res[f"hop{hop}_returned_score_vs_fp64_maxabs"] = round(exact_err, 6)
        res[f"hop{hop}_bruteforce_kth_minus_returned_kth_max"] = round(beaten, 6)
        res[f"hop{hop}_id_agreement_with_bruteforce"] = round(float(same.float().mean()), 6)
        if bf16_rows:
            res[f"hop{hop}_score_vs_fp32_rows_maxrel"] = round(rel32, 6)
        res[f"hop{hop}_ok"] = bool(exact_err <= 1e-3 and beaten <= 2e-3 and bool((same | near).all()) and rel32 <= 1e-2)
    res["full_size_exact"] = bool(res["hop1_ok"] and res["hop2_ok"])
    return res


def _search_case(idx, q, chunks, dup_ok=False):
    """One k = 1 search shape on a structured corpus: ms per call (HIP events, 10 back-to-back calls after 3 warm-ups), which tier decided, the candidate counts, and
    top-1 ids against a brute-force fp32 matmul over all rows (`chunks`: callable c -> rows of chunk c, or a list of resident chunks). An id that differs counts as agreeing
    when the brute force scores the returned row within 2e-3 of its own maximum (exact copies and fp32-matmul noise; reported separately from the strict agreement)."""
    nq = q.shape[0]
    D, I = idx.search_device(q, 1)
    torch.cuda.synchronize()
    t = idx.telemetry(nq, 1)
    for _ in range(3):
        idx.search_device(q, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        idx.search_device(q, 1)
    e1.record()
    torch.cuda.synchronize()
    bs = torch.full((nq,), -float("inf"), device=q.device)
    bi = torch.full((nq,), -1, dtype=torch.int64, device=q.device)
    got = torch.full((nq,), -float("inf"), device=q.device)  # the brute force's own score of the row the index returned
    row0 = 0
    n_chunks = chunks[0] if isinstance(chunks, tuple) else len(chunks)
    for c in range(n_chunks):
        x = chunks[1](c) if isinstance(chunks, tuple) else chunks[c]
        sc = q @ x.T
        s_, a_ = sc.max(1)
        better = s_ > bs
        bs, bi = torch.where(better, s_, bs), torch.where(better, a_ + row0, bi)
        loc = I[:, 0] - row0
        here = (loc >= 0) & (loc < x.shape[0])
        if bool(here.any()):
            got = torch.where(here, sc.gather(1, loc.clamp(0, x.shape[0] - 1)[:, None])[:, 0], got)
        row0 += x.shape[0]
        del x, sc
    strict = float((I[:, 0] == bi).float().mean())
    tie_ok = float(((I[:, 0] == bi) | ((bs - got).abs() <= 2e-3)).float().mean())
    return {"ms_per_search": round(e0.elapsed_time(e1) / 10, 4), "kernel": idx.last_kernel(), "int8_tier_decided": bool(t["i8_tier"] and not t["i8_overflow"]),
            "exact_fallback_ran": bool(t["fallback"]), "candidates_emitted": t["candidates"], "candidates_rescored": t["i8_refined"],
            "top1_id_agreement_with_bruteforce": round(strict, 4), "top1_agreement_up_to_exact_ties": round(tie_ok, 4),
            "returned_score_vs_bruteforce_maxabs": round(float((D[:, 0] - got).abs().max()), 6)}


def _search_case_k(idx, q, chunks, k):
    """A beam > 1 search shape on a structured corpus (the reference's default --beam-size 5 / BASELINE configs[3]'s beam 4): ms per call, whether the screen decided, and
    the returned lists against a brute-force fp32 matmul over all rows (same noise rule as verify_full_size: a differing id is fine inside 2e-3 of score)."""
    nq = q.shape[0]
    D, I = idx.search_device(q, k)
    torch.cuda.synchronize()
    t = idx.telemetry(nq, k)
    for _ in range(2):
        idx.search_device(q, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        idx.search_device(q, k)
    e1.record()
    torch.cuda.synchronize()
    run_s = torch.full((nq, k), -float("inf"), device=q.device)
    run_i = torch.full((nq, k), -1, dtype=torch.int64, device=q.device)
    row0 = 0
    for x in chunks:
        s, i = torch.topk(q @ x.T, k, dim=1)
        cat_s, cat_i = torch.cat([run_s, s], 1), torch.cat([run_i, i + row0], 1)
        run_s, o = torch.topk(cat_s, k, dim=1)
        run_i = torch.gather(cat_i, 1, o)
        row0 += x.shape[0]
    differ = run_i != I
    ok = bool(((run_s - D).abs()[differ] <= 2e-3).all()) and float((run_s[:, k - 1] - D[:, k - 1]).max()) <= 2e-3
    return {"ms_per_search": round(e0.elapsed_time(e1) / 5, 4), "kernel": idx.last_kernel(), "exact_fallback_ran": bool(t["fallback"]),
            "candidates_per_query": round(t["candidates"] / nq, 1), "id_agreement_with_bruteforce": round(float((~differ).float().mean()), 4),
            "lists_equal_up_to_matmul_noise": ok}


def anisotropic_subresult(args, device):
    """The MIPS of the headline (5M x 768, beam 1, 100 / 200 queries per call) on ANISOTROPIC rows -- m u + N(0, 1) for a fixed unit
    vector u, m = args.aniso_m (dense-retrieval embeddings share a large common component; iid rows are the easy case for any
    screening bound): ms per search, which tier decided, candidates, ids against a brute-force fp32 matmul over all rows. The int8
    plane is centred / scaled per column (csrc/mdr_mips.hip col_sum_kernel), so this should read like the iid numbers."""
    from multihop_dense_retrieval_amd import index as mdr_index
    N, d = args.rows, args.dim
    g = torch.Generator(device=device).manual_seed(777)
    u = torch.randn(d, generator=g, device=device)
    u = u / u.norm()
    out = {}
    nch = -(-N // CHUNK_ROWS)
    for m in args.aniso_m:
        idx = mdr_index.IndexFlatIP(d, device=device)
        idx.reserve(N)

        def rows_of(c, m=m):
            return corpus_chunk(0, c, CHUNK_ROWS, d, device)[: min(CHUNK_ROWS, N - c * CHUNK_ROWS)] + m * u
        planted = None
        for c in range(nch):
            x = rows_of(c)
            if c == 0:
                planted = x[torch.arange(200, device=device) * 1009 % x.shape[0]].clone()
            idx.add(x)
            del x
        res = {}
        for nq in (100, 200):
            q = (planted[:nq] + 0.05 * corpus_chunk(5, nq, nq, d, device)).contiguous()
            res[f"nq{nq}"] = _search_case(idx, q, (nch, rows_of))
        out[f"m={m:g}"] = res
        del idx
        torch.cuda.empty_cache()
    return out


def structured_subresult(args, device):
    """the k = 1 search of the headline on rows SHAPED LIKE REAL EMBEDDINGS (scripts/structured_corpora.py), same telemetry as `anisotropic`:
    (i) clustered -- 20 k centres, rows = centre + 0.3 N(0,1), 1 % exact copies; queries = a row + 0.05 noise (first half) and a new cluster member (second half);
    (ii) encoder geometry -- rows = this repo's HIP encoder outputs (random-init roberta-base: shared LayerNorm bias, nearly collapsed rows) for args.rows synthetic
    passages of --geom-len tokens, queries from the same encoder (near-duplicates of corpus passages + fresh sequences)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import structured_corpora as sc
    from multihop_dense_retrieval_amd import index as mdr_index
    N, d = args.rows, args.dim
    nch = -(-N // CHUNK_ROWS)
    out = {}

    def stats(chunks):
        x = chunks[0][:50_000]
        mu = x.mean(0)
        xc = x - mu
        return {"mean_norm": round(float(mu.norm()), 3), "row_norm_mean": round(float(x.norm(dim=1).mean()), 3),
                "centred_row_norm_mean": round(float(xc.norm(dim=1).mean()), 4), "column_std_min_max": [round(float(xc.std(0).min()), 5), round(float(xc.std(0).max()), 5)]}

    if "clustered" in args.structured:
        centres = sc.cluster_centres(device)
        idx = mdr_index.IndexFlatIP(d, device=device)
        idx.reserve(N)
        chunks = []
        for c in range(nch):
            x = sc.clustered_chunk(centres, c, min(CHUNK_ROWS, N - c * CHUNK_ROWS), device)
            idx.add(x)
            chunks.append(x)
        res = {"corpus": f"{N} rows: {sc.N_CENTRES} centres ~ N(0,1), row = centre + {sc.SPREAD} N(0,1), {sc.DUP_FRACTION:.0%} exact copies",
               "queries": "first half: corpus row + 0.05 N(0,1); second half: centre + 0.3 N(0,1)", "stats": stats(chunks)}
        for nq in (100, 200):
            q, _ = sc.clustered_queries(centres, chunks[0], nq, device)
            res[f"nq{nq}"] = _search_case(idx, q, chunks)
        q, _ = sc.clustered_queries(centres, chunks[0], 400, device)
        res["k4_nq400"] = _search_case_k(idx, q, chunks, 4)
        out["clustered"] = res
        del idx, chunks, centres
        torch.cuda.empty_cache()
    if "encoder" in args.structured:
        from multihop_dense_retrieval_amd.retriever import RobertaCtxEncoder
        lo, hi = args.geom_len
        model = RobertaCtxEncoder.random_init(device=device, seed=3)
        idx = mdr_index.IndexFlatIP(d, device=device)
        idx.reserve(N)
        chunks = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _, x in sc.encoder_rows(model, N, lo, hi, device):
            idx.add(x)
            chunks.append(x.clone())
        torch.cuda.synchronize()
        enc_s = time.perf_counter() - t0
        res = {"corpus": f"{N} rows = RobertaCtxEncoder (random-init roberta-base, HIP) embeddings of synthetic passages of {lo}..{hi} tokens",
               "queries": "first half: corpus passages with three tokens changed; second half: fresh sequences; same encoder",
               "encode_seconds": round(enc_s, 1), "stats": stats(chunks)}
        for nq in (100, 200):
            q = sc.encoder_queries(model, nq, lo, hi, device)
            res[f"nq{nq}"] = _search_case(idx, q, chunks)
        res["k4_nq400"] = _search_case_k(idx, sc.encoder_queries(model, 400, lo, hi, device), chunks, 4)
        out["encoder_geometry"] = res
        del idx, chunks, model
        torch.cuda.empty_cache()
    return out


def calls_nq(pipe):
    calls = pipe.search_calls()
    return int(calls[-1][1]) if calls else 0


def mips_roofline(pipe, local, args, d):
    """Roofline entry of the MIPS kernel from the timed search calls of `pipe`: PHYSICAL HBM bytes / HIP-event time of each whole
    search call / 8 TB/s (the fp32-equivalent "algorithmic" rate under its own key)."""
    calls = pipe.search_calls()  # [(ms, nq)] of every timed local search (rank-local)
    shard_bytes = local.stream_bytes()  # N_pad * d * 4 (fp32 index) or * 2 (bf16): the corpus read once
    qpps = [local.queries_per_pass(nq, args.beam) for _, nq in calls]
    passes = [max(1, -(-nq // qpp)) for (_, nq), qpp in zip(calls, qpps)]
    qpp = max(qpps) if qpps else 0
    tot_ms = sum(ms for ms, _ in calls)
    alg_bytes = float(sum(shard_bytes * p for p in passes))
    kname = local.last_kernel()
    ratio, src, fresh = 1.0, "designed (no counter profile for this kernel / shape)", None
    if args.pmc_traffic is None:
        now = src_sha16()
        for name, ent in load_pmc_table().items():
            if name in kname and d == 768 and args.storage != "bf16":
                ratio = float(ent["ratio"])
                fresh = ent.get("csrc_sha16") is not None and ent.get("csrc_sha16") == now
                src = (f"{ent['source']} (measured FETCH_SIZE x 2 / algorithmic bytes = {ratio}; "
                       + ("kernel source unchanged since that measurement" if fresh else
                          f"STALE: csrc/mdr_mips* was {ent.get('csrc_sha16')} when measured, is {now} now -- re-run scripts/measure/gpu_pmc_screen.sh") + ")")
        hbm_bytes = alg_bytes * ratio
    else:
        hbm_bytes, src = float(args.pmc_traffic) * len(calls), "--pmc-traffic"
    n_calls = max(1, len(calls))
    achieved = hbm_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    alg_rate = alg_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": round(hbm_bytes / n_calls), "traffic_source": src, "traffic_fresh": fresh,
            "algorithmic_bytes_per_launch": round(alg_bytes / n_calls),
            "algorithmic_GBps": round(alg_rate, 1),
            "algorithmic_over_physical": round(alg_bytes / hbm_bytes, 3) if hbm_bytes > 0 else None,
            "avg_launch_ms": round(tot_ms / n_calls, 4), "launches_timed": len(calls),
            "queries_per_launch": round(float(np.mean([nq for _, nq in calls])), 1) if calls else 0,
            "corpus_passes_per_launch": round(float(np.mean(passes)), 3) if passes else 0, "queries_per_pass": qpp,
            "note": "achieved/frac = physical HBM bytes (rocprofv3 FETCH_SIZE, gfx950-corrected) / HIP-event time of the whole search call; "
                    "the screen kernels stream only the int8 plane (beam = 1; 776 B per row) or the fp16 hi plane, so the fp32-equivalent (algorithmic) rate is reported separately. "
                    "Pipelined loop: one 256-query pass serves hop 2 of a batch and hop 1 of the next (MFMA-heavier, lower HBM fraction, half "
                    "the passes); `sequential.mips_roofline` is the beam=1, 100-queries-per-call kernel the north star's HBM target names"}


def build_pipeline(args, world, rank, device, dist, weak):
    from multihop_dense_retrieval_amd import index as mdr_index
    from multihop_dense_retrieval_amd import mhop
    N, d, B = args.rows, args.dim, args.batch
    skw = {"storage": "bf16"} if args.storage == "bf16" else {}
    if world > 1:
        sidx = mdr_index.ShardedIndexFlatIP(d, N, local_index=mdr_index.IndexFlatIP(d, device=device, **skw))
        lo, hi = sidx.lo, sidx.hi
        local = sidx.local
    else:
        local = mdr_index.IndexFlatIP(d, device=device, **skw)
        sidx = local
        lo, hi = 0, N
    local.reserve(hi - lo)
    # planted hop-1 answers make the MIPS-only mode self-checking at full size: question i's best row is p_i
    GB = B * world if weak else B  # questions per step over all ranks
    planted = (torch.arange(GB, device=device, dtype=torch.int64) * 48_611 + 17) % N
    kept = build_shard(local, lo, hi, d, device, keep_rows=planted)
    rows_sum = torch.zeros((GB, d), device=device)
    for sel, rows in kept:
        rows_sum[sel] += rows
    if world > 1:
        dist.all_reduce(rows_sum)
    if weak:  # this rank's own questions
        rows_sum, planted = rows_sum[rank * B:(rank + 1) * B].contiguous(), planted[rank * B:(rank + 1) * B]
    return sidx, local, lo, hi, GB, planted, rows_sum


def timed_steps(pipe, args, world, device, dist):
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.step()
    barrier()
    pipe.reset_kernel_timers()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.step()
    barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return out, elapsed


def encode_corpus_mode(args):
    """Secondary measurement: passages/s of the corpus encoder on one GPU -- encode_corpus.predict() driven by
    length-bucketed windows of a synthetic pre-tokenised corpus (title+text pairs of 20..300 tokens, max_c_len 300), embeddings
    copied back and written into a host matrix as the CLI does. Tokenisation itself (host CPU, DataLoader workers) is not part of
    it: there are no BPE files offline and it runs beside the GPU in the CLI."""
    import types

    from multihop_dense_retrieval_amd import encode_corpus
    from multihop_dense_retrieval_amd.retriever import RobertaCtxEncoder
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = RobertaCtxEncoder.random_init(device=device, seed=3)
    n, bs, Lmax = args.passages, args.predict_batch_size, 300
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(20, Lmax + 1, (n,), generator=g)

    offs = torch.zeros(n + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(lens, 0)
    toks = torch.randint(3, 50265, (int(offs[-1]),), generator=g)  # the whole "tokenised corpus", made once
    toks[offs[:-1]] = 0
    toks[offs[1:] - 1] = 2

    class Synth(torch.utils.data.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            ids = toks[offs[i]:offs[i + 1]].view(1, -1)
            return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

    cfg = types.SimpleNamespace(predict_batch_size=bs, num_workers=8, embed_save_path="/tmp/mdr_bench_emb", save_bf16=False, length_bucket_window=16)
    ds = Synth()
    if args.encode_once:
        import shutil
        need = n * 768 * 4 + (1 << 30)
        on_shm = os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 1.2 * need
        cfg.embed_save_path = "/dev/shm/mdr_bench_emb" if on_shm else "/tmp/mdr_bench_emb"
        t0 = time.perf_counter()
        path, done = encode_corpus.encode_shard(model, ds, cfg, 0, 1, 768)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        emb = np.load(path, mmap_mode="r")
        probe = np.asarray(emb[[0, n // 2, n - 1]])
        finite = bool(np.isfinite(probe).all() and (np.abs(probe).sum(1) > 0).all())
        os.remove(path)
        ex, pad = encoder_flops(lens.numpy(), Lmax)
        st = model._lane(0)
        print(json.dumps({"metric": "passages/sec (corpus encoder end to end: encode_shard, RoBERTa-base, max_c_len 300)", "value": round(n / el, 1), "unit": "passages/s",
                          "n_gpus": 1, "passages": n, "rows_written": int(done), "predict_batch_size": bs, "tokens": int(lens.sum()), "seconds": round(el, 2),
                          "executed_TFLOPs": round(ex / el / 1e12, 1), "mfma_frac": round(ex / el / 1e12 / MFMA_PEAK_TFLOPS, 4), "output_on_tmpfs": on_shm,
                          "workers": cfg.num_workers, "host_threads": len(os.sched_getaffinity(0)), "first_mid_last_rows_finite_and_nonzero": finite,
                          "graph_captures": model.graph_captures, "graph_replays": model.graph_replays, "graph_shapes_cached": len(st.graphs),
                          "data": "synthetic pre-tokenised passages (20..300 tokens), length-bucketed windows of 16 batches of 1000",
                          "note": "ONE pass, cold: DataLoader worker start-up, collation + IPC, pinned H2D, forward, D2H, memmap write, final flush"}), flush=True)
        return
    # (a) the GPU side alone: the same length-bucketed batches, collated once and resident on the device; forward + D2H of the
    #     embeddings + write into the host matrix (what predict() does per batch)
    coll = encode_corpus.LengthBucketCollate(bs)
    batches = []
    for w0 in range(0, n, 16 * bs):
        for rows, b in coll([(i, ds[i]) for i in range(w0, min(n, w0 + 16 * bs))]):
            batches.append((rows, encode_corpus.expand_compact(b, lambda t: {k: v.to(device) for k, v in t.items()})))
    out = np.zeros((n, 768), np.float32)

    def gpu_pass():
        torch.cuda.synchronize()
        t = time.perf_counter()
        with torch.no_grad():
            for rows, b in batches:
                out[rows.numpy()] = model(b)["embed"].cpu().numpy()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    gpu_pass()  # sizes the workspace, captures the graph shapes that repeat
    el_gpu = min(gpu_pass(), gpu_pass())
    # (b) end to end through encode_shard(): DataLoader workers (collation + IPC) + pinned H2D + forward + D2H + memmap write
    path, done = encode_corpus.encode_shard(model, ds, cfg, 0, 1, 768)
    t0 = time.perf_counter()
    path, done = encode_corpus.encode_shard(model, ds, cfg, 0, 1, 768)
    torch.cuda.synchronize()
    el2 = time.perf_counter() - t0
    ex, pad = encoder_flops(lens.numpy(), Lmax)
    os.remove(path)
    # (c) the same with the output matrix on tmpfs: what part of (b) is the box's disk (the final msync of n x 768 x 4 bytes)
    el3 = None
    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK):
        cfg.embed_save_path = "/dev/shm/mdr_bench_emb"
        try:
            path3, _ = encode_corpus.encode_shard(model, ds, cfg, 0, 1, 768)
            t0 = time.perf_counter()
            path3, _ = encode_corpus.encode_shard(model, ds, cfg, 0, 1, 768)
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t0
            os.remove(path3)
        except OSError:
            el3 = None
    st = model._lane(0)
    print(json.dumps({"metric": "passages/sec (corpus encoder, RoBERTa-base, max_c_len 300)", "value": round(n / el_gpu, 1), "unit": "passages/s", "n_gpus": 1,
                      "passages": n, "predict_batch_size": bs, "tokens": int(lens.sum()), "seconds": round(el_gpu, 3),
                      "executed_TFLOPs": round(ex / el_gpu / 1e12, 1), "mfma_frac": round(ex / el_gpu / 1e12 / MFMA_PEAK_TFLOPS, 4),
                      "padded_equivalent_TFLOPs": round(pad / el_gpu / 1e12, 1),
                      "end_to_end_with_dataloader": {"value": round(n / el2, 1), "unit": "passages/s", "seconds": round(el2, 3), "workers": cfg.num_workers,
                                                     "host_threads": len(os.sched_getaffinity(0)),
                                                     "output_on_tmpfs": None if el3 is None else {"value": round(n / el3, 1), "seconds": round(el3, 3)},
                                                     "note": "encode_shard(): worker collation + IPC + pinned H2D + forward + D2H + memmap write + final flush of the "
                                                             "matrix to the box's disk; output_on_tmpfs = the same with the matrix in /dev/shm (the DataLoader alone "
                                                             "delivers 55 k passages/s on 8 cores)"},
                      "graph_captures": model.graph_captures, "graph_replays": model.graph_replays, "graph_shapes_cached": len(st.graphs),
                      "data": "synthetic pre-tokenised passages (20..300 tokens), length-bucketed windows of 16 batches of 1000",
                      "note": "value = device-resident token batches -> forward -> D2H of the embeddings -> host matrix (predict()'s per-batch work)"}), flush=True)


def cli_mode(args):
    """queries/s of the DROP-IN CLI: synthetic assets of the headline's size on disk (scripts/cli_bench_assets.py: a 5 M x 768
    fp32 .npy, a 5 M-passage corpus store + token arena, a roberta-base-geometry checkpoint, a real HF byte-level BPE tokenizer over the tiny
    vocabulary, 7 405 questions), then eval_mhop_retrieval.main() with (a) the reference's flags and (b) --hop2-on-device --pipeline-batches. The
    timed region is the CLI's own batch loop ("Encoding questions and searching": tokenisation, both encoder passes, both searches, path ranking,
    metrics, output records; between a device sync on either side), not model / index / corpus loading. The device-resident loop of the default
    mode is timed first in the same process, so the ratio is a same-box number."""
    import shutil

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import cli_bench_assets
    from multihop_dense_retrieval_amd import eval_mhop_retrieval, mhop
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if world > 1 and not args.share_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible (--share-gpu --backend gloo rehearses on one)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    N, d, B = args.rows, args.dim, args.batch
    need = N * (d * 4 + 180 * 4 + 180 * 4 + 64) + (1 << 30)
    out_dir = args.cli_dir
    if out_dir is None:
        shm_ok = os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 1.3 * need
        out_dir = "/dev/shm/mdr_cli_bench" if shm_ok else "/tmp/mdr_cli_bench"
    result = {"metric": "queries/sec (drop-in CLI eval_mhop_retrieval.main, 2-hop, beam-size x topk) over 5Mx768 index", "unit": "queries/s", "n_gpus": world,
              "higher_is_better": True, "data": "synthetic", "dtype": "f32 (index stored as fp16 hi/lo pairs, fp32 accumulate); encoder f16 MFMA / f32 accumulate",
              "config": {"workload": f"scripts/eval/eval_mhop_retrieval.py on synthetic {N}x{d} fp32 index.npy + {N}-passage corpus store, {args.questions} questions, "
                                     f"batch {B}, beam={args.beam} topk={args.topk}, RoBERTa-base geometry (random init), tokenizer = HF byte-level BPE over the tiny vocabulary",
                         "rows": N, "questions": args.questions, "batch": B, "beam": args.beam, "topk": args.topk, "num_workers": args.cli_workers,
                         "host_threads": len(os.sched_getaffinity(0)), "assets_dir": out_dir}}
    # (1) the device-resident loop of the default mode on this box, this process: the number the CLI is compared with
    if world == 1 and not args.no_sequential:
        sidx, local, lo, hi, GB, planted, rows_sum = build_pipeline(args, 1, 0, device, None, False)
        pipe = mhop.SyntheticTwoHop(sidx, batch=B, beam=args.beam, topk=args.topk, dim=d, device=device, max_q_len=args.max_q_len, max_q_sp_len=args.max_q_sp_len,
                                    use_encoder=True, planted_rows=rows_sum, pipelined=True, pool=args.pool)
        _, el = timed_steps(pipe, args, 1, device, None)
        result["device_loop"] = {"value": round(B * args.steps / el, 2), "unit": "queries/s", "ms_per_step": round(el / args.steps * 1e3, 4),
                                 "note": "bench.py default mode (device-resident synthetic loop, software-pipelined), same box, same process"}
        del pipe, sidx, local
        torch.cuda.empty_cache()
    # (2) the assets
    ready = os.path.join(out_dir, "READY")
    if rank == 0:
        reuse = None
        if args.cli_keep and os.path.exists(ready):
            try:
                reuse = json.load(open(ready))
                if reuse.get("rows") != N or reuse.get("questions") != args.questions:
                    reuse = None
            except (OSError, ValueError):
                reuse = None
        for r in range(world):  # markers of an earlier run
            try:
                os.remove(os.path.join(out_dir, f"DONE.rank{r}"))
            except OSError:
                pass
        if reuse is None:
            if os.path.exists(ready):
                os.remove(ready)
            shutil.rmtree(out_dir, ignore_errors=True)
            assets = cli_bench_assets.build(out_dir, N, args.questions, device, log=lambda m: print(m, file=sys.stderr, flush=True))
            assets["rows"], assets["questions"] = N, args.questions
            json.dump(assets, open(ready, "w"))
        else:
            assets = reuse
    else:
        while not os.path.exists(ready):
            time.sleep(0.5)
        assets = json.load(open(ready))
    result["config"]["assets_build_s"] = assets["build_seconds"]
    torch.cuda.empty_cache()
    legs = {"default": [], "device": ["--hop2-on-device"], "unfused": ["--no-pipeline-batches"]}
    base = [assets["raw_data"], assets["indexpath"], assets["corpus_dict"], assets["model_path"], "--batch-size", str(B), "--beam-size", str(args.beam),
            "--topk", str(args.topk), "--shared-encoder", "--model-name", assets["model_name"], "--gpu", "--max-q-len", str(args.max_q_len),
            "--max-q-sp-len", str(args.max_q_sp_len), "--num-workers", str(args.cli_workers)]
    if world > 1:  # every leg is its own W-rank job: each rank's child process inherits RANK / WORLD_SIZE / MASTER_* and opens the process group itself
        base += ["--dist-backend", args.backend] + (["--share-gpu"] if args.share_gpu else [])
    outs = {}
    try:
        import subprocess
        for leg_i, name in enumerate(x for x in args.cli_legs.split(",") if x):
            save = os.path.join(out_dir, f"paths_{name}.jsonl")
            stats = os.path.join(out_dir, f"stats_{name}")
            # a FRESH process per leg, as a user starts the drop-in script (this process has long touched the device: the CLI's worker processes must
            # be forked before that); scripts/gpu_cli_multirank.py = eval_mhop_retrieval.main(argv) + its counters (LAST_RUN) left in a file
            t0 = time.perf_counter()
            env = dict(os.environ)
            if world > 1 and env.get("MASTER_PORT", "").isdigit():
                # every leg's ranks rendezvous on their OWN port, derived the same way on every rank (the legs used to reuse the launcher's
                # port one after another with no settling time between them)
                env["MASTER_PORT"] = str(1024 + (int(env["MASTER_PORT"]) - 1024 + 101 * (leg_i + 1)) % (65535 - 1024))
                # under torchrun the children would otherwise be CLIENTS of the launcher agent's store at the launcher's port (TORCHELASTIC_USE_AGENT_STORE): nobody
                # listens on the derived port and the rendezvous hangs (round 6, first version: the 2-rank leg sat in its 900 s timeout); rank 0 of each leg hosts its own
                env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            rcp = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_cli_multirank.py"), stats] + base + ["--save-path", save] + legs[name],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
            wall = time.perf_counter() - t0
            if rcp.returncode != 0:
                raise SystemExit(f"CLI leg {name} failed:\n{rcp.stderr[-3000:]}")
            print(rcp.stderr[-1500:], file=sys.stderr)
            run = json.load(open(f"{stats}.rank{rank}.json"))
            outs[name] = save
            if rank == 0:
                done = run["stats"].get("batch_done_t", [])
                steady = None
                if len(done) > 12:  # the first batches carry the hipGraph captures: rate over the rest
                    steady = round((len(done) - 1 - 8) * B / (done[-1] - done[8]), 2)
                result[f"cli_{name}"] = {"flags": " ".join(legs[name]) or "(the reference's flags)", "value": round(run["questions"] / run["loop_seconds"], 2),
                                        "unit": "queries/s", "loop_seconds": round(run["loop_seconds"], 4), "ms_per_batch": round(run["loop_seconds"] / max(1, -(-run["questions"] // B)) * 1e3, 4),
                                        "steady_state_queries_per_s": steady, "whole_process_seconds": round(wall, 2), "records": sum(1 for _ in open(save)), "graph_captures": run["graph_captures"],
                                        "graph_replays": run["graph_replays"], "encoder_forward_calls": run["encoder_forward_calls"], "startup_s": run.get("startup_s"),
                                        "stats": {k: v for k, v in run["stats"].items() if k != "batch_done_t"}}
        if rank == 0 and len(outs) >= 2:
            texts = [open(p).read() for p in outs.values()]
            result["legs_jsonl_identical"] = all(t == texts[0] for t in texts[1:])
    finally:
        if world > 1:  # no process group in THIS process (the legs' children own the rendezvous port): rank 0 waits on marker files before it removes the assets
            open(os.path.join(out_dir, f"DONE.rank{rank}"), "w").close()
            deadline = time.time() + 600
            while rank == 0 and time.time() < deadline and not all(os.path.exists(os.path.join(out_dir, f"DONE.rank{r}")) for r in range(world)):
                time.sleep(0.2)
        if rank == 0 and not args.cli_keep:
            shutil.rmtree(out_dir, ignore_errors=True)
    if rank == 0:
        best = max((v for k, v in result.items() if k.startswith("cli_")), key=lambda v: v["value"], default=None)
        if best is not None:
            result["value"] = best["value"]
            result["best_leg"] = best["flags"]
            if "device_loop" in result:
                result["cli_over_device_loop"] = round(best["value"] / result["device_loop"]["value"], 3)
        print(json.dumps(result), flush=True)

def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-exec this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` so that N ranks exist, one per GPU.
    Under a launcher (WORLD_SIZE set) nothing happens here, and a WORLD_SIZE that disagrees with --gpus is an error, not a warning: a scaling record
    whose n_gpus is not what was asked for is void. Fewer than N visible devices is an error too (unless --share-gpu, the one-GPU debugging mode)."""
    if args.mode in ("encode-corpus", "structured"):
        return
    env_world = os.environ.get("WORLD_SIZE")
    under_launcher = env_world is not None and ("RANK" in os.environ or "LOCAL_RANK" in os.environ)
    if env_world is not None and not (int(env_world) == 1 and args.gpus > 1 and not under_launcher):
        # (an environment that merely EXPORTS WORLD_SIZE=1, with no launcher around this process, falls through to the self-launch below: ADVICE r5)
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; refusing to print a line whose n_gpus is not --gpus")
        return
    if args.gpus <= 1:
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    have = torch.cuda.device_count()
    if have < args.gpus and not args.share_gpu:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible (use --share-gpu --backend gloo to rehearse N ranks on one device)")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: self-launching " + " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)  # the launcher sets its own
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // args.gpus)))
    os.execvpe(sys.executable, cmd, env)


def main():
    args = parse()
    self_launch(args)
    if args.mode == "encode-corpus":
        return encode_corpus_mode(args)
    if args.mode == "cli":
        return cli_mode(args)
    if args.mode == "structured":  # the structured-corpus MIPS sub-results alone (profile runs: scripts/measure/r6_structured.sh)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
        torch.cuda.set_device(0)
        print(json.dumps({"metric": "ms per k = 1 search on structured corpora", "rows": args.rows, "geom_len": args.geom_len,
                          "structured": structured_subresult(args, torch.device("cuda", 0))}), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if world > 1 and not args.share_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible (--share-gpu --backend gloo rehearses on one)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    from multihop_dense_retrieval_amd import mhop

    N, d, B = args.rows, args.dim, args.batch
    weak = world > 1 and args.scaling == "weak"
    t0 = time.time()
    sidx, local, lo, hi, GB, planted, rows_sum = build_pipeline(args, world, rank, device, dist, weak)
    torch.cuda.synchronize()
    build_s = time.time() - t0

    pipe = mhop.SyntheticTwoHop(sidx, batch=B, beam=args.beam, topk=args.topk, dim=d, device=device,
                                max_q_len=args.max_q_len, max_q_sp_len=args.max_q_sp_len,
                                use_encoder=not args.no_encoder, planted_rows=rows_sum, rank=rank, world=world, weak=weak,
                                pipelined=not args.sequential, pool=args.pool)
    out, elapsed = timed_steps(pipe, args, world, device, dist)

    # self-checks, outside the timed region: structural properties + exactness against a brute-force pass over ALL rows
    ok = pipe.self_check(out, planted)
    if not args.no_verify:
        ok.update(verify_full_size(out, lo, hi, N, d, device, args.beam, bf16_rows=args.storage == "bf16"))
        if world > 1:  # a shard only sees its own rows: the claim holds when it holds on every rank
            flag = torch.tensor([1.0 if ok["full_size_exact"] else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok["full_size_exact"] = bool(flag.item() > 0.5)

    if args.dump_ids and rank == 0:
        np.savez(args.dump_ids, I=out["I"].cpu().numpy(), I2=out["I2"].cpu().numpy(), D=out["D"].cpu().numpy(), D2=out["D2"].cpu().numpy())

    ms_per_step = elapsed / args.steps * 1e3
    qps = GB * args.steps / elapsed

    roofline = mips_roofline(pipe, local, args, d)

    stage = pipe.stage_ms()
    result = {
        "metric": "queries/sec (2-hop, beam-size x topk) over 5Mx768 index",
        "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
        "dtype": ("bf16 index rows, fp32 accumulate" if args.storage == "bf16" else "f32 (index stored as fp16 hi/lo pairs, fp32 accumulate)")
                 + "; encoder f16 MFMA / f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"synthetic {N}x{d} {'bf16' if args.storage == 'bf16' else 'fp32'} corpus{' row-sharded over ' + str(world) + ' GPUs' if world > 1 else ''}, "
                               f"{B}-question batches{' per GPU (global batch ' + str(GB) + ')' if weak else ''}, 2-hop beam={args.beam} topk={args.topk}"
                               f"{' , MIPS-only (no encoder)' if not pipe.use_encoder else ', RoBERTa-base encoder (random init)'}",
                   "rows": N, "dim": d, "batch": B, "global_batch": GB, "beam": args.beam, "topk": args.topk, "shards": world,
                   "collective_world_size": (dist.get_world_size() if dist is not None else 1), "collective_backend": (dist.get_backend() if dist is not None else None),
                   "encoder": pipe.encoder_desc(), "index_build_s": round(build_s, 2),
                   "loop": getattr(pipe, "loop_desc", None) or (
                       ("software-pipelined: hop 2 of batch i beside hop 1 of batch i+1 = two concurrent encoder forwards (two lanes / streams) + one fused "
                        "corpus pass (every batch still walks the full hop-1 -> hop-2 chain; see `sequential` for the unpipelined loop)") if pipe.pipelined
                       else "sequential: one batch at a time, four dependent stages")},
        "roofline": roofline,
        "self_check": ok,
        "stage_ms": stage,
    }
    if pipe.use_encoder:
        # ---- second roofline entry: the encoder (MFMA-bound), the larger share of the step --------------------------------
        # executed FLOPs averaged over the timed steps (the pool's batches differ): hop 1 from each step's question lengths, hop 2 from the lengths its
        # assembly produced
        log = list(pipe.step_log)
        f1 = [encoder_flops(pipe.batches[ent]["q_len"].cpu().numpy(), args.max_q_len) for ent, _ in log]
        f2 = [encoder_flops(lens.cpu().numpy(), args.max_q_sp_len) for _, lens in log]
        e1, p1 = float(np.mean([f[0] for f in f1])), float(np.mean([f[1] for f in f1]))
        e2, p2 = float(np.mean([f[0] for f in f2])), float(np.mean([f[1] for f in f2]))
        q_lens = np.concatenate([pipe.batches[ent]["q_len"].cpu().numpy() for ent, _ in log]) / 1.0
        sp_lens = np.concatenate([lens.cpu().numpy() for _, lens in log]) / 1.0
        nlog = max(1, len(log))
        enc_ms = stage["hop1_encode"] + stage["hop2_encode"]  # pipelined: hop2_encode = wall time of the hop-2 forward AND the next batch's hop-1 forward beside it (hop1_encode is 0)
        if world > 1 and not weak:  # strong scaling: each rank encodes 1/world of the rows, the rest of the stage is the all-gather
            e1, p1, e2, p2 = e1 / world, p1 / world, e2 / world, p2 / world
        ach = (e1 + e2) / (enc_ms * 1e-3) / 1e12
        result["roofline_encoder"] = {
            "bound": "mfma", "kernel": "mdr_encoder_forward (gemm_big / attention_stream / layernorm kernels, hipGraph replay)",
            "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
            "executed_flop_per_step": round(e1 + e2), "padded_equivalent_flop_per_step": round(p1 + p2),
            "padded_equivalent_TFLOPs": round((p1 + p2) / (enc_ms * 1e-3) / 1e12, 1),
            "hop1": {"ms": stage["hop1_encode"], "tokens": int(q_lens.sum() / nlog),
                     "TFLOPs": round(e1 / (stage["hop1_encode"] * 1e-3) / 1e12, 1) if stage["hop1_encode"] > 0 else None},
            "hop2": {"ms": stage["hop2_encode"], "tokens": int(sp_lens.sum() / nlog),
                     "TFLOPs": round(e2 / (stage["hop2_encode"] * 1e-3) / 1e12, 1),
                     **({"side_stream_hop1": {"tokens": int(q_lens.sum() / nlog), "flop": round(e1),
                                              "note": "the next batch's hop-1 forward runs CONCURRENTLY on a second stream inside this stage's time; "
                                                      "`achieved` above counts both forwards' FLOPs over it, this entry's TFLOPs only the hop-2 forward's"}}
                        if pipe.pipelined else {})},
            "share_of_step": round(enc_ms / ms_per_step, 3),
            "frac_of_sustained_mfma": round(ach / 1700.0, 4),
            "sustained_mfma_note": "secondary: a register-only v_mfma_f32_16x16x32_f16 loop (scripts/ubench/mfma_peak.hip, two waves per SIMD) sustains 1.65-1.77 PFLOP/s "
                                   "on RANDOM operands on this pool (2.35-2.41 on zeros): the chip clocks to its power limit (~1.7 GHz under a full matrix pipe); `frac` above is "
                                   "against the nominal 2.5 PFLOP/s",
            "note": "executed FLOPs (masked tokens dropped, last layer CLS-only) / HIP-event stage time on the launch stream; peak = dense fp16 MFMA"}
    if pipe.use_encoder:
        # ---- the step time over the pool's DIFFERENT batches: min / mean / max, per 1 k hop-2 tokens, and which 256-row tile counts occurred
        ps = pipe.per_step()
        if ps:
            ms = np.array([p[0] for p in ps])
            tk = np.array([p[2] for p in ps], np.float64)
            tiles = {}
            for t in tk:
                key = str(int(-(-t // 256)))
                tiles[key] = tiles.get(key, 0) + 1
            per_k = ms / (tk / 1e3)
            result["per_step"] = {"pool": pipe.pool, "steps": len(ps), "ms": {"min": round(float(ms.min()), 4), "mean": round(float(ms.mean()), 4), "max": round(float(ms.max()), 4)},
                                  "max_over_min_ms": round(float(ms.max() / ms.min()), 4),
                                  "hop2_tokens": {"min": int(tk.min()), "mean": int(tk.mean()), "max": int(tk.max())},
                                  "ms_per_1k_hop2_tokens": {"min": round(float(per_k.min()), 5), "mean": round(float(per_k.mean()), 5), "max": round(float(per_k.max()), 5)},
                                  "max_over_min_ms_per_token": round(float(per_k.max() / per_k.min()), 4),
                                  "row_tiles_of_256_histogram": dict(sorted(tiles.items())),
                                  "note": "one entry per timed step but the last (a step = first event of the step to the first event of the next, main stream); the work of a step "
                                          "scales with its hop-2 tokens, so the tile-round steps of the persistent GEMMs show in ms_per_1k_hop2_tokens"}
    if hasattr(local, "telemetry") and calls_nq(pipe):  # which screening tier decided the LAST search of the timed region (test hook; outside it)
        t = local.telemetry(calls_nq(pipe), args.beam)
        result["mips_tiers"] = {"int8_tier_ran": t["i8_tier"], "int8_tier_handed_over": t["i8_overflow"], "exact_fallback_ran": bool(t["fallback"]),
                                "candidates_emitted_last_group": t["candidates"], "candidates_rescored": t["i8_refined"] if t["i8_tier"] else t["candidates"],
                                "kernel": local.last_kernel()}
    result["stage_share"] = {"encoder": round((stage.get("hop1_encode", 0) + stage.get("hop2_encode", 0)) / ms_per_step, 3),
                             "mips": round((stage.get("hop1_search", 0) + stage.get("hop2_search", 0)) / ms_per_step, 3)}

    if pipe.pipelined and not args.no_sequential:
        # the same job, one batch at a time (the loop exactly as the reference writes it): sub-result of the same line
        mhop.SyntheticTwoHop._defer_encoder = True  # share the encoder and the arena of the main pipeline
        pipe_q = mhop.SyntheticTwoHop(sidx, batch=B, beam=args.beam, topk=args.topk, dim=d, device=device, max_q_len=args.max_q_len,
                                      max_q_sp_len=args.max_q_sp_len, use_encoder=not args.no_encoder, planted_rows=rows_sum, rank=rank,
                                      world=world, weak=weak, pipelined=False, pool=args.pool)
        mhop.SyntheticTwoHop._defer_encoder = False
        if pipe.use_encoder:
            pipe_q.encoder, pipe_q.arena = pipe.encoder, pipe.arena  # same weights, same token arena
        _, el_q = timed_steps(pipe_q, args, world, device, dist)
        result["sequential"] = {"value": round(GB * args.steps / el_q, 2), "unit": "queries/s", "ms_per_step": round(el_q / args.steps * 1e3, 4),
                                "stage_ms": pipe_q.stage_ms(), "mips_roofline": mips_roofline(pipe_q, local, args, d)}
        del pipe_q

    if pipe.pipelined and pipe.use_encoder and world == 1 and not args.no_sequential:
        # the same job in the encoder's OTHER residual-stream numerics modes (mdr_encoder_config.residual_fp32; the headline runs the default, an
        # apex-O1-faithful one: DESIGN.md section 4): what each costs and how far its embeddings are from the default's -- sub-results of the same line
        from multihop_dense_retrieval_amd.retriever import RobertaRetriever
        names = {0: "fp16 residual copy (one more rounding per LayerNorm than apex O1; NOT O1-faithful)",
                 1: "fp32 residual stream, fp32 Linear sums (O1-faithful, more accurate than apex O1)",
                 2: "fp32 residual stream, out-projection / FFN2 outputs rounded to fp16 (apex O1's own dataflow)"}
        result["numerics_mode"] = {"residual_fp32": int(pipe.encoder.residual_fp32), "meaning": names[int(pipe.encoder.residual_fp32)]}
        result["numerics_modes"] = {}
        for mode in (0, 1, 2):
            if mode == int(pipe.encoder.residual_fp32):
                continue
            prev = os.environ.get("MDR_RESIDUAL_FP32")
            os.environ["MDR_RESIDUAL_FP32"] = str(mode)  # read when the encoder object is made (the device handle is created with the mode)
            try:
                enc_m = RobertaRetriever.random_init(device=device, seed=3)
            finally:
                if prev is None:
                    del os.environ["MDR_RESIDUAL_FP32"]
                else:
                    os.environ["MDR_RESIDUAL_FP32"] = prev
            assert int(enc_m.residual_fp32) == mode
            mhop.SyntheticTwoHop._defer_encoder = True
            pipe_r = mhop.SyntheticTwoHop(sidx, batch=B, beam=args.beam, topk=args.topk, dim=d, device=device, max_q_len=args.max_q_len,
                                          max_q_sp_len=args.max_q_sp_len, use_encoder=True, planted_rows=rows_sum, rank=rank, world=world, weak=weak,
                                          pipelined=True, pool=args.pool)
            mhop.SyntheticTwoHop._defer_encoder = False
            pipe_r.encoder, pipe_r.arena = enc_m, pipe.arena
            out_r, el_r = timed_steps(pipe_r, args, world, device, dist)
            result["numerics_modes"][f"residual_fp32={mode}"] = {
                "value": round(GB * args.steps / el_r, 2), "unit": "queries/s", "ms_per_step": round(el_r / args.steps * 1e3, 4), "meaning": names[mode],
                "hop1_embedding_max_abs_diff_vs_default": round(float((out_r["q"] - out["q"]).abs().max()), 6) if out_r["q"].shape == out["q"].shape else None,
                "hop1_ids_equal_to_default": bool((out_r["I"] == out["I"]).all()) if out_r["I"].shape == out["I"].shape else None}
            del pipe_r, enc_m

    if world > 1 and weak and not args.no_strong:
        # The same job with ONE 100-question batch shared by all ranks (the reference's fixed --batch-size): a sub-result of
        # the same JSON line, so a scaling run carries the strong-scaling number next to the weak one.
        mhop.SyntheticTwoHop._defer_encoder = True
        pipe_s = mhop.SyntheticTwoHop(sidx, batch=B, beam=args.beam, topk=args.topk, dim=d, device=device, max_q_len=args.max_q_len,
                                      max_q_sp_len=args.max_q_sp_len, use_encoder=not args.no_encoder,
                                      planted_rows=torch.zeros((B, d), device=device), rank=rank, world=world, weak=False,
                                      pipelined=pipe.pipelined, pool=args.pool)
        mhop.SyntheticTwoHop._defer_encoder = False
        if pipe.use_encoder:
            pipe_s.encoder, pipe_s.arena = pipe.encoder, pipe.arena
        _, el_s = timed_steps(pipe_s, args, world, device, dist)
        result["strong_scaling"] = {"value": round(B * args.steps / el_s, 2), "unit": "queries/s", "ms_per_step": round(el_s / args.steps * 1e3, 4),
                                    "global_batch": B, "note": "one batch of --batch questions for all ranks: encoder slices split over ranks + all-gather"}

    # (part of the full default line only: diagnostic invocations -- --no-cpu-baseline / --no-verify / --no-encoder -- skip it)
    if (rank == 0 and world == 1 and not args.no_anisotropic and args.beam == 1 and args.storage != "bf16" and d == 768
            and not (args.no_cpu_baseline or args.no_verify or args.no_encoder)):
        result["anisotropic"] = anisotropic_subresult(args, device)
        if args.structured:
            result["structured"] = structured_subresult(args, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, device, gpu_index=local)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, device, gpu_index=None):
    """FAISS-equivalent CPU path (kind 'port': faiss is not installed here) on the GPU box's host cores, over ALL rows of the same
    synthetic corpus, no extrapolation: the corpus is re-generated chunk by chunk (250 k rows; the device RNG
    that built the index), each chunk is searched on the host by the two searches of one 100-question step (hop 1 + hop 2: the same
    shape, 100 queries, k = beam), the running top-k is merged exactly as FAISS merges its row blocks, and only the HOST search time
    is summed (generation and the D2H copy of a chunk are not the baseline's work: FAISS holds the matrix in RAM). Threads = the
    cores this container may use (affinity and cgroup quota: the boxes give 16 of the host's 256 hardware threads; 128 OpenMP
    threads on that quota ran 9x slower). Two restatements of the same algorithm are timed on the first chunk and the faster one
    runs the rest: oracle/flat_ip_blas.py (OpenBLAS sgemm behind numpy + top-k, what FAISS itself does) and oracle/flat_ip_oracle.c.
    --cpu-seconds bounds the repetitions per chunk, never the rows."""
    import ctypes
    import subprocess

    from oracle import flat_ip_blas
    cores = flat_ip_blas.usable_cpus()
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    f = lib.mdr_oracle_flat_ip_search
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_int]
    d, B, k = args.dim, args.batch, args.beam
    q_dev = corpus_chunk(1, 0, B, d, device)
    q = q_dev.cpu().numpy()

    def run_c(xb):
        D = np.empty((B, k), np.float32)
        I = np.empty((B, k), np.int64)
        t = time.perf_counter()
        rc = f(q.ctypes.data, B, xb.ctypes.data, xb.shape[0], d, k, D.ctypes.data, I.ctypes.data, cores)
        assert rc == 0
        return time.perf_counter() - t, D, I

    def run_blas(xb):
        t = time.perf_counter()
        Db, Ib = flat_ip_blas.search(q, xb, k)
        return time.perf_counter() - t, Db, Ib

    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cores)
    except Exception:
        limiter = None
    nchunk = -(-args.rows // CHUNK_ROWS)
    first = corpus_chunk(0, 0, CHUNK_ROWS, d, device)[:min(CHUNK_ROWS, args.rows)].cpu().numpy()
    run_c(first), run_blas(first)  # warm both (thread pools, page faults)
    t_probe = {"blas": min(run_blas(first)[0], run_blas(first)[0]), "c_openmp": min(run_c(first)[0], run_c(first)[0])}
    agree = float((run_c(first)[2] == run_blas(first)[2]).mean())
    if t_probe["blas"] <= t_probe["c_openmp"]:
        run, impl = run_blas, "oracle/flat_ip_blas.py (numpy/OpenBLAS sgemm + top-k, the FAISS algorithm)"
    else:
        run, impl = run_c, "oracle/flat_ip_oracle.c (restatement of faiss IndexFlatIP::search, OpenMP)"
    # repetitions of the (hop 1 + hop 2) pair per chunk so that the whole baseline is about --cpu-seconds of host work
    reps = int(max(1, min(5, args.cpu_seconds / max(2 * min(t_probe.values()) * nchunk, 1e-3))))
    run_D = np.full((B, k), -np.inf, np.float32)
    run_I = np.full((B, k), -1, np.int64)
    t_pair_sum, t_all = 0.0, time.perf_counter()
    for c in range(nchunk):
        base = c * CHUNK_ROWS
        n_here = min(CHUNK_ROWS, args.rows - base)
        xb = (first if c == 0 else corpus_chunk(0, c, CHUNK_ROWS, d, device)[:n_here].cpu().numpy())
        best = None
        for _ in range(reps):  # the step's two searches; the fastest repetition of the pair counts (shared host)
            t1, Dc, Ic = run(xb)
            t2, _, _ = run(xb)
            best = t1 + t2 if best is None else min(best, t1 + t2)
        t0 = time.perf_counter()  # merge this block's top-k into the running one (part of a flat search over all rows)
        cat_D, cat_I = np.concatenate([run_D, Dc], 1), np.concatenate([run_I, Ic + base], 1)
        o = np.argsort(-cat_D, 1, kind="stable")[:, :k]
        run_D, run_I = np.take_along_axis(cat_D, o, 1), np.take_along_axis(cat_I, o, 1)
        t_pair_sum += best + 2 * (time.perf_counter() - t0)
        del xb
    wall = time.perf_counter() - t_all
    if limiter is not None and hasattr(limiter, "restore_original_limits"):
        limiter.restore_original_limits()
    check = None
    if gpu_index is not None:  # the same queries through the HIP index: the baseline searched the same matrix and found the same rows
        _, Ig = gpu_index.search_device(q_dev.contiguous(), k)
        check = round(float((Ig.cpu().numpy() == run_I).mean()), 6)
    return {"value": round(B / t_pair_sum, 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"ALL {args.rows} rows of the same synthetic corpus, streamed in {nchunk} chunks of {CHUNK_ROWS} (no extrapolation): per chunk the 2 "
                      f"flat-IP searches of one step (hop 1 + hop 2, {B} queries, k={k}), fastest of {reps} repetition(s), + the running top-k merge; "
                      f"{t_pair_sum:.2f} s of host search time for one step ({wall:.1f} s wall incl. re-generating and copying the chunks) on {cores} threads "
                      f"(the container's CPU quota; host has {os.cpu_count()} hardware threads); MIPS only (the reference's encoder runs on the GPU in the "
                      f"reference too)",
            "impl": impl, "one_search_over_first_chunk_s": {k_: round(v, 4) for k_, v in t_probe.items()}, "probe_id_agreement": agree,
            "rows_searched": int(args.rows), "extrapolated": False, "top1_id_agreement_with_hip_index": check}


if __name__ == "__main__":
    main()

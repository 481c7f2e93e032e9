"""Synthetic corpora with the STRUCTURE of real dense-retrieval embeddings, for the MIPS measurements and property tests.

The reference searches `wiki_index.npy` (/root/reference/scripts/eval/eval_mhop_retrieval.py:93-125,155,179): 5.2 M LayerNorm outputs of one trained
encoder -- clustered, near-duplicate-dense, with a large shared component. None of that is in i.i.d. N(0, 1) rows, which are the easy case for a screening
bound. Two generators, both chunk-keyed (any chunk is reproducible on its own, on the device):

  * clustered:         20 k centres ~ N(0, 1); row = centre[a] + 0.3 N(0, 1); 1 % of the rows are exact copies of another row of the same chunk.
                       Queries: first half = a corpus row + 0.05 N(0, 1) (a clear winner, or an exact tie with its copy), second half = a centre +
                       0.3 N(0, 1) (a new member of a cluster: ~250 rows within a few score units of the winner).
  * encoder geometry:  rows = the HIP encoder's own outputs (RobertaCtxEncoder, random-init roberta-base geometry) for synthetic passages: a shared
                       LayerNorm bias, nearly collapsed rows. Queries = outputs of the same encoder for fresh sequences and for corpus passages with a
                       few tokens changed.

Used by bench.py (`structured` sub-result), tests/test_mips_fullsize_gpu.py and scripts/measure/r6_structured.sh. Needs a device; nothing here is on the product path."""
import torch

D = 768
CHUNK_ROWS = 250_000
N_CENTRES = 20_000
SPREAD = 0.3
DUP_FRACTION = 0.01


def _gen(device, *key):
    s = 0
    for k in key:
        s = s * 1_000_003 + int(k)
    return torch.Generator(device=device).manual_seed(s % (2 ** 62))


def cluster_centres(device, seed=4242):
    return torch.randn((N_CENTRES, D), generator=_gen(device, seed, 1), device=device)


def clustered_chunk(centres, c, rows, device, seed=4242):
    """Rows [c * CHUNK_ROWS, c * CHUNK_ROWS + rows) of the clustered corpus."""
    g = _gen(device, seed, 2, c)
    a = torch.randint(0, centres.shape[0], (rows,), generator=g, device=device)
    x = centres[a] + SPREAD * torch.randn((rows, D), generator=g, device=device)
    ndup = int(rows * DUP_FRACTION)
    if ndup:
        perm = torch.randperm(rows, generator=g, device=device)
        dst, src = perm[:ndup], perm[ndup:2 * ndup]  # disjoint: a copy is never itself copied
        x[dst] = x[src]
    return x


def clustered_queries(centres, chunk0, nq, device, seed=4242):
    """-> (q [nq, D], planted_local_row [nq // 2]): the first half are rows `planted` of chunk 0 + 0.05 noise."""
    g = _gen(device, seed, 3, nq)
    half = nq // 2
    planted = (torch.arange(half, device=device) * 1009 + 7) % chunk0.shape[0]
    q = torch.empty((nq, D), device=device)
    q[:half] = chunk0[planted] + 0.05 * torch.randn((half, D), generator=g, device=device)
    a = torch.randint(0, centres.shape[0], (nq - half,), generator=g, device=device)
    q[half:] = centres[a] + SPREAD * torch.randn((nq - half, D), generator=g, device=device)
    return q.contiguous(), planted


def synthetic_token_batch(c, rows, len_lo, len_hi, device, seed=99):
    """[rows, len_hi] token ids (<s> first, </s> last real token, pad 1 behind) + mask for passages c * rows ...: lengths ~ U[len_lo, len_hi]."""
    g = _gen(device, seed, 5, c)
    lens = torch.randint(len_lo, len_hi + 1, (rows,), generator=g, device=device)
    ids = torch.randint(3, 50265, (rows, len_hi), generator=g, device=device)
    pos = torch.arange(len_hi, device=device)[None, :]
    mask = pos < lens[:, None]
    ids = torch.where(mask, ids, torch.ones_like(ids))
    ids[:, 0] = 0
    ids[torch.arange(rows, device=device), lens - 1] = 2
    return ids, mask.to(torch.int64), lens


def encoder_rows(model, n_rows, len_lo, len_hi, device, per_call=None, sink=None):
    """Yield (row0, x [b, D] fp32) blocks of the encoder-geometry corpus: the model's embeddings of synthetic passages, `per_call` passages per forward
    (default: as many as the encoder's token bound allows). Deterministic in (n_rows, len_lo, len_hi, per_call)."""
    cap = getattr(model, "MAX_TOKENS_PER_CALL", 1 << 17)
    per_call = per_call or max(1, min(8192, cap // len_hi))
    row0, c = 0, 0
    with torch.no_grad():
        while row0 < n_rows:
            b = min(per_call, n_rows - row0)
            ids, mask, _ = synthetic_token_batch(c, per_call, len_lo, len_hi, device)
            x = model({"input_ids": ids[:b], "input_mask": mask[:b]})["embed"].float()
            yield row0, x
            row0 += b
            c += 1


def encoder_queries(model, nq, len_lo, len_hi, device, per_call=None):
    """First half: corpus passages 0 .. nq/2-1 with three tokens changed (near-duplicates of rows); second half: fresh sequences."""
    cap = getattr(model, "MAX_TOKENS_PER_CALL", 1 << 17)
    per_call = per_call or max(1, min(8192, cap // len_hi))
    half = nq // 2
    ids, mask, lens = synthetic_token_batch(0, per_call, len_lo, len_hi, device)
    ids, mask, lens = ids[:half].clone(), mask[:half], lens[:half]
    g = _gen(device, 31337, nq)
    for _ in range(3):
        p = 1 + (torch.randint(0, 1 << 30, (half,), generator=g, device=device) % (lens - 2).clamp(min=1))
        ids[torch.arange(half, device=device), p] = torch.randint(3, 50265, (half,), generator=g, device=device)
    ids2, mask2, _ = synthetic_token_batch(10_000_019, nq - half, len_lo, len_hi, device)
    with torch.no_grad():
        qa = model({"input_ids": ids, "input_mask": mask})["embed"].float()
        qb = model({"input_ids": ids2, "input_mask": mask2})["embed"].float()
    return torch.cat([qa, qb], 0).contiguous()

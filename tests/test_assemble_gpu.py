"""GPU (-m gpu): mdr_assemble_hop2 against a literal Python statement of RoBERTa pair encoding with HF `longest_first`
truncation (transformers 2.11 `truncate_sequences`: remove one token at a time from the longer sequence, from the pair on
ties) and of the empty-passage rule of eval_mhop_retrieval.py:162-165. Integer work: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def pair_reference(q, d, max_len, bos=0, eos=2, pad=1):
    q, d = list(q), list(d)
    for _ in range(max(0, len(q) + len(d) + 4 - max_len)):
        if len(q) > len(d):
            q.pop()
        else:
            d.pop()
    ids = [bos] + q + [eos, eos] + d + [eos]
    mask = [1] * len(ids) + [0] * (max_len - len(ids))
    return ids + [pad] * (max_len - len(ids)), mask


@pytest.mark.parametrize("Lq,out_len,beam", [(70, 350, 1), (70, 350, 4), (12, 40, 3), (30, 24, 2), (8, 9, 1)])
def test_assemble_matches_pair_encoding(Lq, out_len, beam):
    from multihop_dense_retrieval_amd.arena import TokenArena
    rng = np.random.default_rng(Lq * 1000 + out_len)
    n_docs, B = 57, 9
    doc_lens = rng.integers(0, 2 * out_len, n_docs)
    doc_lens[:4] = [0, 1, out_len, out_len - 4]
    docs = [rng.integers(3, 50265, n).astype(np.int32) for n in doc_lens]
    offsets = np.zeros(n_docs + 1, np.int64)
    offsets[1:] = np.cumsum(doc_lens)
    empty = (rng.random(n_docs) < 0.2).astype(np.uint8)
    arena = TokenArena(torch.from_numpy(np.concatenate(docs)), torch.from_numpy(offsets), torch.from_numpy(empty)).to("cuda")
    q_lens = rng.integers(2, Lq + 1, B)  # incl. <s> and </s>
    q_lens[0], q_lens[1] = 2, Lq
    q_ids = np.full((B, Lq), 1, np.int64)
    q_mask = np.zeros((B, Lq), np.int64)
    for b, n in enumerate(q_lens):
        q_ids[b, :n] = rng.integers(3, 50265, n)
        q_ids[b, 0], q_ids[b, n - 1] = 0, 2
        q_mask[b, :n] = 1
    doc_ids = rng.integers(0, n_docs, (B, beam)).astype(np.int64)
    doc_ids[0, 0], doc_ids[1, 0] = 2, 0
    D = rng.standard_normal((B, beam)).astype(np.float32)
    Dt = torch.from_numpy(D.copy()).cuda()
    ids, mask = arena.assemble_hop2(torch.from_numpy(q_ids).cuda(), torch.from_numpy(q_mask).cuda(), torch.from_numpy(doc_ids).cuda(), Dt, out_len)
    ids, mask, Dn = ids.cpu().numpy(), mask.cpu().numpy(), Dt.cpu().numpy()
    for b in range(B):
        for j in range(beam):
            e_ids, e_mask = pair_reference(q_ids[b, 1:q_lens[b] - 1], docs[doc_ids[b, j]], out_len)
            assert list(ids[b * beam + j]) == e_ids, (b, j)
            assert list(mask[b * beam + j]) == e_mask, (b, j)
            assert Dn[b, j] == (-np.inf if empty[doc_ids[b, j]] else D[b, j])


def test_arena_from_corpus_and_roundtrip(tmp_path):
    from multihop_dense_retrieval_amd.arena import TokenArena

    class Tok:
        def __call__(self, text, add_special_tokens=False):  # a text or, like the HF tokenizers, a batch of texts
            if isinstance(text, (list, tuple)):
                return {"input_ids": [[3 + len(w) for w in t.split()] for t in text]}
            return {"input_ids": [3 + len(w) for w in text.split()]}
    id2doc = {"0": {"title": "A b", "text": "x yy zzz"}, "1": {"title": "Only title here", "text": "  "}, "2": {"title": "T", "text": "q"}}
    a = TokenArena.from_corpus(id2doc, Tok())
    assert a.offsets.tolist() == [0, 3, 6, 7] and a.empty.tolist() == [0, 1, 0] and a.tokens.tolist() == [4, 5, 6, 7, 8, 7, 4]
    a.save(tmp_path / "arena.npz")
    b = TokenArena.load(tmp_path / "arena.npz").to("cuda")  # load() maps the members; .to() sends them through mdr_upload_host
    assert torch.equal(a.tokens, b.tokens.cpu()) and torch.equal(a.offsets, b.offsets.cpu()) and torch.equal(a.empty, b.empty.cpu())
    big = TokenArena(torch.arange(5_000_011, dtype=torch.int32) % 50265, torch.tensor([0, 5_000_011]), None)  # > 4 MiB: the pinned double-buffer path
    big.save(tmp_path / "big.npz")
    c = TokenArena.load(tmp_path / "big.npz").to("cuda")
    assert c.tokens.dtype == torch.int32 and torch.equal(c.tokens.cpu(), big.tokens) and c.empty is None

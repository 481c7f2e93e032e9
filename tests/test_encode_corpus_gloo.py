"""CPU (-m "not gpu"), world_size 2 over gloo: the multi-rank branch of the corpus encoder that replaces the reference's
DataParallel wrap (/root/reference/scripts/encode_corpus.py:85-89) -- contiguous row split, ONE memory-mapped .npy shared
by the ranks, id2doc.json written once by rank 0, length-bucketed windows that still put every passage in its own row.
The HIP encoder is replaced by a deterministic stand-in (a function of the token ids only), because what is under test
here is the host orchestration; the same `encode_shard` runs under RCCL with the real RobertaCtxEncoder on the GPU box
(tests/test_cli_gpu.py runs it single-rank, tests/test_rccl_multi_gpu.py with two ranks when two GPUs are visible)."""
import argparse
import json
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytest.importorskip("transformers")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HID = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_tokenizer():
    import transformers
    bpe = os.path.join(ROOT, "tests", "golden", "tiny_bpe")
    vocab = json.load(open(os.path.join(bpe, "vocab.json")))
    merges = [tuple(ln.split()) for ln in open(os.path.join(bpe, "merges.txt")).read().split("\n") if ln and not ln.startswith("#")]
    return transformers.RobertaTokenizer(vocab=vocab, merges=merges)


class StandInEncoder:
    """embed = f(own tokens): padding- and batch-composition-invariant, so any row split / batching must reproduce it."""

    def eval(self):
        return self

    def __call__(self, batch):
        ids, m = batch["input_ids"].double(), batch["input_mask"].double()
        pos = torch.arange(ids.shape[1], dtype=torch.float64)[None, :] + 1.0
        feats = [m.sum(1), (ids * m).sum(1), (ids * m * pos).sum(1), (ids * ids * m).sum(1) % 9973.0]
        feats += [ids[:, min(j, ids.shape[1] - 1)] * m[:, min(j, ids.shape[1] - 1)] for j in range(1, 5)]
        return {"embed": torch.stack(feats, 1).float()}


def expected_rows(tok, docs, max_c_len):
    from multihop_dense_retrieval_amd import data
    enc = StandInEncoder()
    rows = []
    for d in docs:
        text = d["text"].strip() if d["text"].strip() else d["title"]
        ids, mask = data.encode_pairs_2_11(tok, [data.normalize(d["title"].strip())], [text], max_c_len, False)
        rows.append(enc({"input_ids": torch.tensor(ids), "input_mask": torch.tensor(mask)})["embed"][0].numpy())
    return np.stack(rows)


def make_docs(n, seed=0):
    rng = np.random.RandomState(seed)
    words = "the quick brown fox retrieval encoder index beam passage question answer Paris London film band album city".split()
    docs = [{"title": f"Title {i}", "text": " ".join(rng.choice(words, rng.randint(1, 60)))} for i in range(n)]
    docs[3]["text"] = "  "  # empty passage -> title
    return docs


def _worker(rank, world, port, corpus, save, bs, window, workers):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multihop_dense_retrieval_amd import data, encode_corpus
    tok = make_tokenizer()
    args = argparse.Namespace(predict_batch_size=bs, num_workers=workers, embed_save_path=save, save_bf16=True, length_bucket_window=window)
    ds = data.EmDataset(tok, corpus, 70, 40, False, save, write_id2doc=(rank == 0))
    path, done = encode_corpus.encode_shard(StandInEncoder(), ds, args, rank, world, HID, barrier=dist.barrier, to_device=lambda b: b)
    lo, hi = encode_corpus.shard_range(len(ds), world, rank)
    assert done == hi - lo
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bs,window,workers", [(53, 8, 3, 0), (40, 16, 1, 0), (7, 4, 16, 2)])
def test_two_ranks_fill_one_shared_matrix(tmp_path, n, bs, window, workers):
    docs = make_docs(n, seed=n)
    corpus = tmp_path / "corpus.jsonl"
    corpus.write_text("".join(json.dumps(d) + "\n" for d in docs))
    save = str(tmp_path / "emb")
    mp.spawn(_worker, args=(2, _free_port(), str(corpus), save, bs, window, workers), nprocs=2, join=True)
    xb = np.load(save + ".npy")
    assert xb.shape == (n, HID) and xb.dtype == np.float32
    want = expected_rows(make_tokenizer(), docs, 40)
    assert np.array_equal(xb, want)  # every passage in its own row, whichever rank / window / batch encoded it
    side = np.load(save + ".bf16.npy")
    assert np.array_equal(side, torch.from_numpy(xb).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
    id2doc = json.load(open(os.path.join(save, "id2doc.json")))  # written once, by rank 0, complete
    assert len(id2doc) == n and id2doc["3"] == [docs[3]["title"], docs[3]["text"], False]


def test_shard_range_is_a_partition():
    from multihop_dense_retrieval_amd.encode_corpus import shard_range
    from multihop_dense_retrieval_amd.index import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                lo, hi = shard_range(n, w, r)
                assert (lo, hi) == shard_bounds(n, w, r)
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def test_length_bucket_collate_keeps_row_indices():
    from multihop_dense_retrieval_amd.encode_corpus import LengthBucketCollate
    samples = [(10 + i, {"input_ids": torch.arange(L)[None, :], "attention_mask": torch.ones((1, L), dtype=torch.int64)})
               for i, L in enumerate([5, 2, 9, 2, 7, 1])]
    from multihop_dense_retrieval_amd.data import em_collate
    from multihop_dense_retrieval_amd.encode_corpus import expand_compact
    out = LengthBucketCollate(4)(samples)
    assert [r.tolist() for r, _ in out] == [[15, 11, 13, 10], [14, 12]]  # sorted by (length, row), cut into batches of 4
    # workers hand over compact batches (one int32 token vector + lengths); expand_compact rebuilds em_collate's dict bit for bit
    assert all("_compact_flat" in b and b["_compact_flat"].dtype == torch.int32 for _, b in out)
    by_row = {r: s for r, s in samples}
    for rows, b in out:
        got = expand_compact(b, lambda x: x)
        want = em_collate([by_row[r] for r in rows.tolist()])
        assert sorted(got) == sorted(want) and all(torch.equal(got[k], want[k]) and got[k].dtype == want[k].dtype for k in want)
    b0 = expand_compact(out[0][1], lambda x: x)
    assert b0["input_ids"].shape == (4, 5) and expand_compact(out[1][1], lambda x: x)["input_ids"].shape == (2, 9)
    assert b0["input_mask"].sum(1).tolist() == [1, 2, 2, 5]
    # a mask that is not all ones (or token_type_ids) keeps the padded form
    odd = [(0, {"input_ids": torch.arange(3)[None, :], "attention_mask": torch.tensor([[1, 1, 0]])})]
    assert "input_ids" in LengthBucketCollate(4)(odd)[0][1]

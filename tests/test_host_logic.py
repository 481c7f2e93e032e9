"""CPU (-m "not gpu"): the product's host-side logic (multihop_dense_retrieval_amd/mhop.py, data.py, config.py)
against the frozen restatement of the reference's inline code (tests/golden/mhop.json, from
oracle/mhop_oracle.py) and against outputs of the reference's own functions (tests/golden/collate.npz)."""
import json

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from multihop_dense_retrieval_amd import config, data, mhop  # noqa: E402
from oracle import mhop_oracle  # noqa: E402


def _cases(golden):
    g = golden("mhop.json")
    return g, mhop.load_corpus_dict(dict(g["id2doc_list"]))


def test_strip_and_corpus_dict(golden):
    g, id2doc = _cases(golden)
    assert [mhop.strip_question(it["question"]) for it in g["items"]] == g["cases"][0]["stripped"]
    assert mhop.strip_question("a??") == "a?" and mhop.strip_question("a") == "a"
    assert id2doc["1"] == {"title": "Beta", "text": "beta text"}
    already = {"0": {"title": "T", "text": "x", "extra": 1}}
    assert mhop.load_corpus_dict(already) is already  # dict-valued corpora pass through untouched (extra keys kept)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_two_hop_host_logic_matches_frozen_reference_restatement(golden, ci):
    g, id2doc = _cases(golden)
    c = g["cases"][ci]
    beam, topk = c["beam"], c["topk"]
    D, I = np.array(c["D"], np.float32), np.array(c["I"], np.int64)
    D2, I2 = np.array(c["D2"], np.float32), np.array(c["I2"], np.int64)
    pairs = mhop.build_hop2_pairs(c["stripped"], D, I, id2doc)
    assert [list(p) for p in pairs] == c["pairs"]
    exp_after = np.array([[-np.inf if v is None else v for v in row] for row in c["D_after"]], np.float32)
    assert np.array_equal(D, exp_after)  # empty text -> title, hop-1 score -inf, in place
    chains = mhop.rank_paths(D, I, D2, I2, beam, topk)
    exp = [[(h1, h2, -np.inf if s is None else s) for h1, h2, s in ch] for ch in c["chains"]]
    for got, want in zip(chains, exp):
        assert [(a, b) for a, b, _ in got] == [(a, b) for a, b, _ in want]
        assert np.allclose([s for *_, s in got], [s for *_, s in want], equal_nan=True)
    metrics, lines = [], []
    for it, ch, wm, wj in zip(g["items"], chains, c["metrics"], c["jsonl"]):
        m = mhop.question_metrics(ch, it["sp"], id2doc)
        m.update(question=it["question"], type=it["type"])
        assert m == wm
        metrics.append(m)
        assert json.dumps(mhop.output_record(it, ch, id2doc)) == wj  # byte-identical JSONL line
    assert mhop.summary_lines(metrics) == c["log"]


def test_rank_paths_agrees_with_oracle_on_random_inputs_and_device_variant():
    rng = np.random.default_rng(0)
    for beam, topk in [(1, 1), (4, 4), (5, 2), (8, 64)]:
        B = 7
        D = -np.sort(-rng.standard_normal((B, beam)).astype(np.float32), 1)
        I = rng.integers(0, 1000, (B, beam))
        D2 = -np.sort(-rng.standard_normal((B * beam, beam)).astype(np.float32), 1)
        I2 = rng.integers(0, 1000, (B * beam, beam))
        a = mhop.rank_paths(D, I, D2, I2, beam, topk)
        b = mhop_oracle.rank_paths(D, I, D2, I2, beam, topk)
        assert [[(x, y) for x, y, _ in ch] for ch in a] == [[(x, y) for x, y, _ in ch] for ch in b]
        h1, h2, s = mhop.rank_paths_device(torch.from_numpy(D), torch.from_numpy(I), torch.from_numpy(D2), torch.from_numpy(I2), beam, topk)
        assert [[(int(x), int(y)) for x, y in zip(r1, r2)] for r1, r2 in zip(h1, h2)] == [[(x, y) for x, y, _ in ch] for ch in a]
    with pytest.raises(IndexError):  # topk > beam^2, as in the reference (eval_mhop_retrieval.py:197-198)
        mhop.rank_paths(D[:, :1], I[:, :1], D2[:B, :1], I2[:B, :1], 1, 2)


def test_metrics_assert_on_bad_sp(golden):
    _, id2doc = _cases(golden)
    with pytest.raises(AssertionError):
        mhop.question_metrics([(0, 1, 0.0)], ["Alpha", "Alpha"], id2doc)


def test_collate_matches_reference_functions(golden):
    g = golden("collate.npz")
    vals = [torch.arange(10, 10 + n, dtype=torch.long) for n in g["lens"]]
    assert np.array_equal(data.collate_tokens(vals, 0).numpy(), g["pad0"])
    assert np.array_equal(data.collate_tokens(vals, 1).numpy(), g["pad1"])
    assert np.array_equal(data.collate_tokens(vals, 1, left_pad=True).numpy(), g["left"])
    samples = [{"input_ids": v.view(1, -1), "attention_mask": torch.ones(1, len(v), dtype=torch.long)} for v in vals]
    b = data.em_collate(samples)
    assert np.array_equal(b["input_ids"].numpy(), g["em.input_ids"]) and np.array_equal(b["input_mask"].numpy(), g["em.input_mask"])
    assert data.em_collate([]) == {}


def test_encode_args_flags():
    a = config.encode_args(["--do_predict", "--predict_batch_size", "1000", "--model_name", "roberta-base", "--predict_file", "c.jsonl",
                            "--init_checkpoint", "m.pt", "--embed_save_path", "out", "--fp16", "--max_c_len", "300", "--num_workers", "20"])
    assert (a.predict_batch_size, a.max_c_len, a.embed_save_path, a.fp16, a.is_query_embed, a.local_rank) == (1000, 300, "out", True, False, -1)
    assert a.max_q_len == 50 and a.fp16_opt_level == "O1" and a.shared_encoder is False


def test_em_dataset_writes_id2doc_and_pairs_title_text(tmp_path):
    class RobertaToy:  # class name contains "Roberta" -> empty text falls back to the title (encode_datasets.py:89-91)
        def __call__(self, a, text_pair=None, max_length=None, truncation=None, return_tensors=None):
            ids = [0] + [3 + len(w) for w in a.split()] + [2, 2] + [3 + len(w) for w in text_pair.split()] + [2]
            ids = ids[:max_length]
            return {"input_ids": torch.tensor([ids]), "attention_mask": torch.ones(1, len(ids), dtype=torch.long)}
    corpus = tmp_path / "c.jsonl"
    corpus.write_text("\n".join(json.dumps(d) for d in [{"title": "A b", "text": "x yy zzz"}, {"title": "Empty", "text": " "},
                                                        {"title": "I", "text": "t", "intro": True}]))
    ds = data.EmDataset(RobertaToy(), str(corpus), 20, 6, False, str(tmp_path / "emb"))
    assert json.load(open(tmp_path / "emb" / "id2doc.json")) == {"0": ["A b", "x yy zzz", False], "1": ["Empty", " ", False], "2": ["I", "t", True]}
    assert len(ds) == 3 and ds[0]["input_ids"].shape[1] == 6  # truncated to max_c_len
    assert ds[1]["input_ids"].tolist() == [[0, 8, 2, 2, 8, 2]]  # title used as text
    b = data.em_collate([ds[0], ds[2]])
    assert b["input_ids"].shape == (2, 6) and b["input_mask"].sum().item() == 6 + ds[2]["input_ids"].shape[1]

"""CPU (-m "not gpu"): the product's host-side logic (multihop_dense_retrieval_amd/mhop.py, data.py, config.py)
against the frozen restatement of the reference's inline code (tests/golden/mhop.json, from
oracle/mhop_oracle.py) and against outputs of the reference's own functions (tests/golden/collate.npz)."""
import json
import os

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
        bos_token_id, eos_token_id, pad_token_id = 0, 2, 1

        def __call__(self, texts, add_special_tokens=True, truncation=False):  # the surface data.encode_pairs_2_11 uses: bare BPE of a list
            assert add_special_tokens is False and truncation is False
            return {"input_ids": [[3 + len(w) for w in t.split()] for t in texts]}
    corpus = tmp_path / "c.jsonl"
    corpus.write_text("\n".join(json.dumps(d) for d in [{"title": "A b", "text": "x yy zzz"}, {"title": "Empty", "text": " "},
                                                        {"title": "I", "text": "t", "intro": True}]))
    ds = data.EmDataset(RobertaToy(), str(corpus), 20, 6, False, str(tmp_path / "emb"))
    assert json.load(open(tmp_path / "emb" / "id2doc.json")) == {"0": ["A b", "x yy zzz", False], "1": ["Empty", " ", False], "2": ["I", "t", True]}
    assert len(ds) == 3 and ds[0]["input_ids"].tolist() == [[0, 4, 2, 2, 4, 2]]  # longest-first truncation to max_c_len = 6 (title "A b" and text lose one token each... the pop loop: 2+3 -> 1+1)
    assert ds[1]["input_ids"].tolist() == [[0, 8, 2, 2, 8, 2]]  # title used as text
    b = data.em_collate([ds[0], ds[2]])
    assert b["input_ids"].shape == (2, 6) and b["input_mask"].sum().item() == 6 + ds[2]["input_ids"].shape[1]


def test_answer_recall_matches_reference_functions(golden):
    """--only-eval-ans: SimpleTokenizer words, para_has_answer, the chain concatenation (no separator between chains) and
    the log lines, against outputs of the reference's own functions (tests/golden/answer_recall.json)."""
    import unicodedata

    from multihop_dense_retrieval_amd import answer_recall as ar
    g = golden("answer_recall.json")
    for para, toks in zip(g["paras"], g["tokens"]):
        assert ar.simple_words(unicodedata.normalize("NFD", para)) == toks
    for i, para in enumerate(g["paras"]):
        for j, ans in enumerate(g["answers"]):
            assert ar.para_has_answer(ans, para) == g["has_answer"][i][j], (para, ans)
    metrics = []
    for it, chains, concat, want in zip(g["items"], g["chains"], g["concat"], g["metrics"]):
        ch = [(int(a), int(b), 0.0) for a, b in chains]
        assert ar.chain_text(ch, g["id2doc"]) == concat
        m = ar.answer_metrics(it, ch, g["id2doc"])
        assert m == want
        metrics.append(m)
    assert ar.answer_summary_lines(metrics) == g["log"]
    with pytest.raises(AssertionError):
        ar.para_has_answer("not a list", "text")


def test_rank_paths_with_separate_beam_widths_matches_the_fever_expression():
    """eval_mhop_fever.py:130-150: D_ reshaped to [B, beam1, beam2], path = D[:, :, None] + D_, reversed argsort of the
    ravel, unravel over (beam1, beam2) -- restated literally here (the script cannot be imported: top-level faiss/apex)."""
    rng = np.random.default_rng(9)
    B, b1, b2, topk = 4, 2, 5, 7
    D = -np.sort(-rng.standard_normal((B, b1)).astype(np.float32), axis=1)
    D[1, 1] = -np.inf
    I = rng.integers(0, 50, (B, b1)).astype(np.int64)
    D2 = -np.sort(-rng.standard_normal((B * b1, b2)).astype(np.float32), axis=1)
    I2 = rng.integers(0, 50, (B * b1, b2)).astype(np.int64)
    got = mhop.rank_paths(D, I, D2, I2, b1, topk, beam2=b2)
    D_, I_ = D2.reshape(B, b1, b2), I2.reshape(B, b1, b2)
    path_scores = np.expand_dims(D, axis=2) + D_
    for idx in range(B):
        ranked = np.vstack(np.unravel_index(np.argsort(path_scores[idx].ravel())[::-1], (b1, b2))).transpose()
        want = [(int(I[idx, p[0]]), int(I_[idx, p[0], p[1]])) for p in ranked[:topk]]
        assert [(h1, h2) for h1, h2, _ in got[idx]] == want
    with pytest.raises(IndexError):
        mhop.rank_paths(D, I, D2, I2, b1, b1 * b2 + 1, beam2=b2)


def test_fever_record_looks_text_up_by_title():
    from multihop_dense_retrieval_amd import eval_mhop_fever as fv
    id2doc = {"0": ["A", "first a", True], "1": ["B", "b text", False], "2": ["A", "second a", False]}
    title2doc = {item[0]: item[1] for item in id2doc.values()}  # later duplicates win, like the reference (:80)
    rec = fv.fever_record({"id": 7, "claim": "c", "label": "x"}, [(0, 1, 0.5), (2, 0, 0.1)], id2doc, title2doc)
    assert list(rec.keys()) == ["id", "claim", "candidate_chains"]
    assert json.loads(json.dumps(rec))["candidate_chains"] == [[["A", "second a"], ["B", "b text"]], [["A", "second a"], ["A", "second a"]]]
    a = fv.build_parser().parse_args(["d", "i", "c", "m"])
    assert (a.topk, a.max_q_len, a.max_q_sp_len, a.beam_size_1, a.beam_size_2, a.batch_size, a.model_name) == (2, 45, 400, 5, 5, 100, "bert-base-uncased")


def test_corpus_store_round_trips_the_corpus_dict(tmp_path):
    """The memory-mapped store answers like the parsed JSON dict: same keys, same {"title","text"} values, both the dict and
    the [title, text, intro] list forms, unicode, empty texts; json.dumps of an entry is what the reference would write."""
    from multihop_dense_retrieval_amd import corpus_store
    docs = {str(i): {"title": f"T{i} \u00e9", "text": ("" if i == 3 else f"text {i} \u6771\u4eac " * (i % 4))} for i in range(37)}
    src = tmp_path / "id2doc.json"
    src.write_text(json.dumps(docs))
    st = corpus_store.CorpusStore(corpus_store.build_store(str(src), str(tmp_path / "id2doc.json.store")))
    assert len(st) == 37 and list(st)[:3] == ["0", "1", "2"] and "36" in st and "37" not in st and "x" not in st and "07" not in st
    for k, v in docs.items():
        assert st[k] == v and json.dumps(st[k]) == json.dumps(v)
    assert mhop.load_corpus_dict(str(tmp_path / "id2doc.json.store"))["5"]["title"] == docs["5"]["title"]
    lists = {str(i): [f"t{i}", f"x{i}", i % 2 == 0] for i in range(5)}
    st2 = corpus_store.CorpusStore(corpus_store.build_store(lists, str(tmp_path / "l.store")))
    assert st2.as_list("2") == ["t2", "x2", True] and st2["1"] == {"title": "t1", "text": "x1", "intro": False}
    assert {v["title"]: v["text"] for v in st2.values()} == {f"t{i}": f"x{i}" for i in range(5)}
    with pytest.raises(ValueError):
        corpus_store.build_store({"0": ["a", "b"], "2": ["c", "d"]}, str(tmp_path / "bad.store"))


def test_int8_screen_bound_is_rigorous():
    """The inequality the int8 screening tier of csrc/mdr_mips.hip rests on (restated in numpy, float64 as ground truth):
    x = s (x8 + e), q = t (q8 + f), |e|, |f| <= 1/2  =>  |q.x - t s sum(q8 x8)| <= t s (L1(q8)/2 + L1(x8)/2 + d/4),
    with the quantisation written the way convert_to_i8_kernel / prep_queries_i8_kernel write it (fp32 scale, rintf, clamp)."""
    rng = np.random.RandomState(0)
    d = 768

    def quant(v):
        v = v.astype(np.float32)
        mx = np.abs(v).max(axis=1, keepdims=True)
        sc = np.where(mx > 0, mx / np.float32(127), np.float32(0)).astype(np.float32)
        inv = np.where(mx > 0, np.float32(127) / mx, np.float32(0)).astype(np.float32)
        v8 = np.clip(np.rint(v * inv), -127, 127).astype(np.int64)
        return v8, sc[:, 0].astype(np.float64)

    cases = {
        "gauss": (rng.randn(300, d), rng.randn(40, d)),
        "mean": (rng.randn(300, d) + 30.0, rng.randn(40, d) + 5.0),
        "heavy": (rng.standard_cauchy((300, d)), rng.standard_cauchy((40, d))),
        "tiny": (rng.randn(300, d) * 1e-6, rng.randn(40, d) * 1e6),
        "sparse": (rng.randn(300, d) * (rng.rand(300, d) < 0.01), rng.randn(40, d) * (rng.rand(40, d) < 0.05)),
    }
    for name, (x, q) in cases.items():
        x = np.clip(x, -3e4, 3e4)
        x8, s = quant(x)
        q8, t = quant(q)
        true = q.astype(np.float32).astype(np.float64) @ x.astype(np.float32).astype(np.float64).T
        approx = (q8 @ x8.T) * t[:, None] * s[None, :]
        bound = t[:, None] * s[None, :] * (0.5 * np.abs(q8).sum(1)[:, None] + 0.5 * np.abs(x8).sum(1)[None, :] + 0.25 * d)
        assert (np.abs(true - approx) <= bound * 1.001 + 1e-300).all(), name
        if name == "gauss":  # ... and it is not vacuous: a few tenths of the score spread
            assert np.median(bound) < 1.0 * true.std()
    # The plane is CENTRED (round 3): it stores x - c for a fixed vector c (fp32 subtraction), the bounds are bounds on q.(x - c), and an
    # exact score e = q.x enters the centred domain as e - (q.c + slack), slack = 1e-4 sum|q_i c_i| + 2e-6 |q.c| (prep_queries_i8_kernel).
    # Checked: (a) the interval still holds for the centred vectors, (b) e - (fl32(q.c) + slack) is a LOWER bound of the true centred
    # score (so `known` / the refinement thresholds never cut a winner), (c) for rows with a large common component -- one outlier
    # coordinate, a dense common mean -- centring shrinks the bound by about the size of that component.
    for name, common in {"outlier-coordinate": np.eye(1, d, 7)[0] * 40.0, "dense-mean": np.full(d, 200.0 / np.sqrt(d))}.items():
        x = rng.randn(300, d)
        if name == "outlier-coordinate":
            x[:, 7] *= 0.1  # large and nearly constant, as in real embeddings
        x = (x + common).astype(np.float32)
        q = (rng.randn(40, d) + (1.0 if name == "outlier-coordinate" else 0.2) * common).astype(np.float32)
        c = x[:200].mean(0, dtype=np.float64).astype(np.float32)   # column means of the first rows, as add() takes them
        sd = x[:200].std(0)
        ref = np.sqrt((sd ** 2).mean())   # centre_finish_kernel: w_i in [A_i / X, X / B_i], 1 where possible, a power of two
        A, B = sd / ref, (np.abs(c) + 3.5 * sd) / (3.5 * ref)
        X = np.sqrt(max(1.0, (A * B).max()))
        wr = np.where(A / X > 1, A / X, np.where(X / B < 1, X / B, 1.0))
        w = np.exp2(np.clip(np.rint(np.log2(np.maximum(wr, 1e-30))), -12, 12)).astype(np.float32)
        xc = ((x - c) / w).astype(np.float32)                      # fp32 subtraction + exact scaling, as convert_to_i8_kernel does it
        x8, s = quant(xc)
        q8, t = quant((q * w).astype(np.float32))                  # the query enters as q_i w_i: sum (q_i w_i) ((x_i - c_i) / w_i) = q.(x - c)
        true_c = q.astype(np.float64) @ (x.astype(np.float64) - c.astype(np.float64)).T
        approx = (q8 @ x8.T) * t[:, None] * s[None, :]
        bound = t[:, None] * s[None, :] * (0.5 * np.abs(q8).sum(1)[:, None] + 0.5 * np.abs(x8).sum(1)[None, :] + 0.25 * d)
        assert (np.abs(true_c - approx) <= bound * 1.001).all(), name
        qc32 = np.zeros(len(q), np.float32)
        for j in range(d):  # a sequential fp32 sum: the worst order the device could use
            qc32 = (qc32 + q[:, j] * c[j]).astype(np.float32)
        slack = 1e-4 * (np.abs(q.astype(np.float64)) * np.abs(c.astype(np.float64))).sum(1) + 2e-6 * np.abs(qc32)
        exact = q.astype(np.float64) @ x.astype(np.float64).T
        exact32 = exact * (1 + 3e-6 * np.sign(rng.randn(*exact.shape)))  # the fp32 re-scoring's own relative error (exact_dot16)
        assert ((exact32 - (qc32.astype(np.float64) + slack)[:, None]) <= true_c + 1e-3 * bound).all(), name
        x8u, su = quant(x)
        q8u, tu = quant(q)
        bound_u = tu[:, None] * su[None, :] * (0.5 * np.abs(q8u).sum(1)[:, None] + 0.5 * np.abs(x8u).sum(1)[None, :] + 0.25 * d)
        assert np.median(bound_u) > 3.0 * np.median(bound), (name, np.median(bound_u), np.median(bound))


def test_arena_cache_is_rebuilt_when_its_tokenisation_rule_differs(tmp_path, tiny_roberta_tokenizer):
    """<corpus_dict>.arena.npz carries a tag (tokenisation rule, tokenizer class, vocabulary hash, empty-text rule, token cap); a cache written
    under another tag -- or by a revision that wrote no tag -- loads as None, so the CLI rebuilds it instead of mixing two rules in one hop-2 input."""
    import numpy as np
    from multihop_dense_retrieval_amd import data
    from multihop_dense_retrieval_amd.arena import TokenArena, arena_tag
    tok = tiny_roberta_tokenizer
    id2doc = {"0": {"title": "T0", "text": "the title and text"}, "1": {"title": "the born", "text": "  "}}
    a = TokenArena.from_corpus(id2doc, tok, roberta=True, max_tokens=40)
    tag = arena_tag(tok, True, 40)
    path = str(tmp_path / "c.json.arena.npz")
    a.save(path, tag=tag)
    b = TokenArena.load(path, expect_tag=tag)
    assert b is not None and np.array_equal(np.asarray(b.tokens), a.tokens.numpy()) and np.array_equal(np.asarray(b.offsets), a.offsets.numpy()) and b.empty.tolist() == [0, 1]  # (load(): memory-mapped members)
    assert TokenArena.load(path, expect_tag=arena_tag(tok, True, 350)) is None          # another token cap
    assert TokenArena.load(path, expect_tag=arena_tag(tok, False, 40)) is None          # another empty-text rule
    old = data.PREFIX_SPACE_2_11
    try:
        data.PREFIX_SPACE_2_11 = False
        assert arena_tag(tok, True, 40) != tag and TokenArena.load(path, expect_tag=arena_tag(tok, True, 40)) is None  # the rule ADVICE r3 named
    finally:
        data.PREFIX_SPACE_2_11 = old
    np.savez(str(tmp_path / "old.npz"), tokens=a.tokens.numpy(), offsets=a.offsets.numpy(), empty=a.empty.numpy())  # a cache of round 3: no tag
    assert TokenArena.load(str(tmp_path / "old.npz"), expect_tag=tag) is None
    assert TokenArena.load(str(tmp_path / "old.npz")) is not None  # (explicit loads without a tag still work)


def test_wait_for_file_deadline_heartbeat_and_failure_marker(tmp_path, caplog):
    """ranks != 0 poll for files rank 0 builds (corpus store, token arena) -- with a deadline, a heartbeat and a failure marker."""
    from multihop_dense_retrieval_amd import eval_mhop_retrieval as ev
    target = str(tmp_path / "thing")
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ev.wait_for_file(lambda: False, target, "the thing", timeout=0.3, poll=0.05, beat=0.1)
    with pytest.raises(ValueError):
        with ev._failure_marker(target):
            raise ValueError("disk full")
    with pytest.raises(RuntimeError, match="rank 0 failed to build the thing.*disk full"):
        ev.wait_for_file(lambda: False, target, "the thing", timeout=5, poll=0.05)
    os.utime(target + ".failed", (1.0, 1.0))  # a marker from an earlier job (older than this process) is ignored: the wait runs into its deadline instead
    with pytest.raises(RuntimeError, match="gave up waiting"):
        ev.wait_for_file(lambda: False, target, "the thing", timeout=0.2, poll=0.05)
    with ev._failure_marker(target):  # a later successful build clears the marker
        pass
    assert not os.path.exists(target + ".failed")
    ev.wait_for_file(lambda: True, target, "the thing", timeout=1)


def test_index_numerics_mismatch_warns(tmp_path, caplog):
    import types
    from multihop_dense_retrieval_amd import eval_mhop_retrieval as ev
    idx = str(tmp_path / "emb.npy")
    assert ev.check_index_numerics(idx, types.SimpleNamespace(residual_fp32=2)) is None  # no metadata, no check
    with open(ev.index_meta_path(idx), "w") as f:
        json.dump({"encoder_numerics": {"residual_fp32": 0}}, f)
    import logging
    with caplog.at_level(logging.WARNING):
        assert ev.check_index_numerics(idx, types.SimpleNamespace(residual_fp32=2)) == 0
    assert "residual_fp32=0" in caplog.text and "MDR_RESIDUAL_FP32=0" in caplog.text


def test_arena_npz_members_are_memory_mapped_and_round_trip(tmp_path):
    """TokenArena.load maps the members of the uncompressed .npz (no 3.6 GB read at 5 M passages); the mapped arrays equal what np.load returns,
    a stale tag still returns None, and a compressed file falls back to np.load."""
    from multihop_dense_retrieval_amd import arena
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 40, 500)
    offs = np.zeros(501, np.int64)
    offs[1:] = np.cumsum(lens)
    toks = rng.integers(3, 500, int(offs[-1])).astype(np.int32)
    empty = (rng.random(500) < 0.1).astype(np.uint8)
    a = arena.TokenArena(torch.from_numpy(toks), torch.from_numpy(offs), torch.from_numpy(empty))
    p = str(tmp_path / "c.arena.npz")
    a.save(p, tag="T1")
    z = arena._npz_memmap(p)
    assert isinstance(z["tokens"], np.memmap) and isinstance(z["offsets"], np.memmap) and str(z["tag"]) == "T1"
    ref = np.load(p)
    for k in ("tokens", "offsets", "empty"):
        assert np.array_equal(np.asarray(z[k]), ref[k]) and z[k].dtype == ref[k].dtype
    b = arena.TokenArena.load(p, expect_tag="T1")
    assert b.n_docs == 500 and not torch.is_tensor(b.tokens) and np.array_equal(np.asarray(b.tokens), toks)
    assert arena.TokenArena.load(p, expect_tag="T2") is None
    c = b.to("cpu")  # (the device path, through mdr_upload_host, is exercised by tests/test_assemble_gpu.py)
    assert torch.is_tensor(c.tokens) and c.tokens.dtype == torch.int32 and np.array_equal(c.tokens.numpy(), toks) and np.array_equal(c.empty.numpy(), empty)
    b.save(str(tmp_path / "again.npz"), tag="T1")  # a lazily loaded arena can be written back
    assert np.array_equal(np.load(str(tmp_path / "again.npz"))["tokens"], toks)
    a2 = arena.TokenArena(torch.from_numpy(toks), torch.from_numpy(offs), None)
    a2.save(str(tmp_path / "noempty.npz"), tag="T1")
    assert arena.TokenArena.load(str(tmp_path / "noempty.npz"), expect_tag="T1").empty is None
    np.savez_compressed(str(tmp_path / "z.npz"), tokens=toks, offsets=offs, empty=empty, tag=np.array("T1"))
    assert arena._npz_memmap(str(tmp_path / "z.npz")) is None
    d = arena.TokenArena.load(str(tmp_path / "z.npz"), expect_tag="T1")
    assert torch.is_tensor(d.tokens) and np.array_equal(d.tokens.numpy(), toks)

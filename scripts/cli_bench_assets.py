"""Synthetic assets for `bench.py --mode cli`: everything the drop-in CLI (scripts/eval/eval_mhop_retrieval.py) reads, at the
headline's size, built without a network in about a minute:

  qas.json                 N_Q questions (8..40 words), two gold titles each, alternating bridge / comparison
  index.npy                ROWS x 768 fp32, the same counter-keyed N(0, 1) rows as bench.py's device-resident corpus
  corpus.store             the memory-mapped corpus store (corpus_store.py layout) of ROWS passages "T<i>" / 60..300 words
  corpus.store.arena.npz   the token arena of those passages (what --hop2-on-device tokenises once), tagged for the tokenizer below
  q_encoder.pt             a `module.`-prefixed state dict of roberta-base geometry, seeded random weights
  roberta-base-synthetic/  config.json (roberta-base geometry) + a REAL HF byte-level BPE RobertaTokenizer over tests/golden/tiny_bpe
                           (the roberta-base vocabulary does not exist offline)

The passages are made of words that the tiny vocabulary holds as ONE space-prefixed token each, so the text blob, the arena and the
tokenizer agree by construction (checked on a sample): the host path tokenises real text with the real tokenizer class, the device path
reads the same tokens from the arena, and both assemble the same hop-2 inputs.
"""
import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BPE = os.path.join(ROOT, "tests", "golden", "tiny_bpe")
CHUNK_ROWS = 250_000  # bench.py's corpus chunk (same RNG keys -> same matrix)


def make_tokenizer():
    import transformers
    with open(os.path.join(BPE, "vocab.json")) as f:
        vocab = json.load(f)
    with open(os.path.join(BPE, "merges.txt")) as f:
        merges = [tuple(ln.split()) for ln in f.read().split("\n") if ln and not ln.startswith("#")]
    return transformers.RobertaTokenizer(vocab=vocab, merges=merges)


def word_table(tok, width=3):
    """Words of `width` letters that tokenise to exactly one token after a space: (list of words, their token ids)."""
    vocab = tok.get_vocab()
    words = sorted(k[1:] for k in vocab if k.startswith("Ġ") and len(k) == width + 1 and k[1:].isalpha() and k[1:].isascii())
    keep = [w for w in words if tok(" " + w, add_special_tokens=False)["input_ids"] == [vocab["Ġ" + w]]]
    assert len(keep) >= 8, "tiny vocabulary has too few whole-word tokens"
    return keep, np.array([vocab["Ġ" + w] for w in keep], np.int32)


def build(out_dir, rows, n_q, device, seed=11, min_len=60, max_len=300, log=print):
    from multihop_dense_retrieval_amd import corpus_store
    from multihop_dense_retrieval_amd.arena import TokenArena, arena_tag
    from multihop_dense_retrieval_amd.retriever import RobertaConfig, expected_state_dict_shapes
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    tok = make_tokenizer()
    words, word_ids = word_table(tok)
    W = len(words)
    width = len(words[0])
    model_dir = os.path.join(out_dir, "roberta-base-synthetic")
    os.makedirs(model_dir, exist_ok=True)
    tok.save_pretrained(model_dir)
    import transformers
    transformers.RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, max_position_embeddings=514,
                               type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2).save_pretrained(model_dir)

    # ---- passages: lengths and word indices on the device, text blob and arena from the same draw -------------------------------
    g = torch.Generator(device=device).manual_seed(seed)
    lens = torch.randint(min_len, max_len + 1, (rows,), generator=g, device=device)
    offs = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1].item())
    widx = torch.randint(0, W, (total,), generator=g, device=device, dtype=torch.int32)
    arena_tokens = torch.from_numpy(word_ids).to(device)[widx.long()]
    arena = TokenArena(arena_tokens, offs, None)
    arena_path = os.path.join(out_dir, "corpus.store.arena.npz")
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(model_dir)  # the object the CLI will load: its tag is the one the cache must carry
    arena.save(arena_path, tag=arena_tag(tok, True, 350))
    log(f"[assets] arena {total} tokens {time.time() - t0:.1f}s")
    # text blob: every word as " www" (the leading space is 2.11's prefix space for the first word): text i = blob[(w+1) offs[i] : (w+1) offs[i+1]]
    table = np.frombuffer("".join(" " + w for w in words).encode(), np.uint8).reshape(W, width + 1)
    widx_h = widx.cpu().numpy()
    del widx, arena_tokens
    titles = [b"T%d" % i for i in range(rows)]
    tlen = np.fromiter((len(t) for t in titles), np.int64, rows)
    title_off = np.zeros(rows + 1, np.int64)
    title_off[1:] = np.cumsum(tlen)
    text_off = offs.cpu().numpy() * (width + 1) + title_off[rows]
    store = os.path.join(out_dir, "corpus.store")
    with open(store, "wb") as f:
        f.write(corpus_store.MAGIC)
        f.write(np.array([rows, int(text_off[rows])], np.int64).tobytes())
        f.write(title_off.tobytes())
        f.write(text_off.tobytes())
        f.write(np.zeros(rows, np.uint8).tobytes())
        f.write(b"".join(titles))
        step = 1 << 24
        flat = table.reshape(-1).view(np.uint32) if width + 1 == 4 else None  # 4-byte words: a 1-D take instead of a row gather
        for lo in range(0, total, step):
            f.write((flat[widx_h[lo:lo + step]] if flat is not None else table[widx_h[lo:lo + step]]).tobytes())
    del widx_h
    log(f"[assets] corpus store {os.path.getsize(store) / 1e9:.2f} GB {time.time() - t0:.1f}s")
    # the three representations agree (sample)
    cs = corpus_store.CorpusStore(store)
    offs_h = offs.cpu().numpy()
    for i in (0, 1, rows // 2, rows - 1):
        doc = cs[str(i)]
        ids = tok(doc["text"], add_special_tokens=False)["input_ids"]
        want = arena.tokens[offs_h[i]:offs_h[i + 1]].cpu().tolist()
        assert doc["title"] == f"T{i}" and ids == want, (i, len(ids), len(want))
    del arena

    # ---- index: the bench's synthetic matrix as a .npy --------------------------------------------------------------------------
    index_path = os.path.join(out_dir, "index.npy")
    mm = np.lib.format.open_memmap(index_path, mode="w+", dtype=np.float32, shape=(rows, 768))
    for c in range(-(-rows // CHUNK_ROWS)):
        gc = torch.Generator(device=device).manual_seed(0 * 1_000_003 + c)
        blk = torch.randn((CHUNK_ROWS, 768), generator=gc, device=device, dtype=torch.float32)
        n = min(CHUNK_ROWS, rows - c * CHUNK_ROWS)
        mm[c * CHUNK_ROWS:c * CHUNK_ROWS + n] = blk[:n].cpu().numpy()
    mm.flush()
    del mm
    log(f"[assets] index.npy {os.path.getsize(index_path) / 1e9:.2f} GB {time.time() - t0:.1f}s")

    # ---- checkpoint ---------------------------------------------------------------------------------------------------------------
    gs = torch.Generator(device=device).manual_seed(3)
    sd = {}
    for k, shp in expected_state_dict_shapes(RobertaConfig()).items():
        if k.endswith("LayerNorm.weight") or k == "project.1.weight":
            t = 1.0 + 0.1 * torch.randn(shp, generator=gs, device=device)
        elif k.endswith("bias"):
            t = 0.1 * torch.randn(shp, generator=gs, device=device)
        elif "embeddings" in k:
            t = 0.5 * torch.randn(shp, generator=gs, device=device)
        else:
            t = (1.5 / shp[1] ** 0.5) * torch.randn(shp, generator=gs, device=device)
        sd["module." + k] = t.cpu()
    ckpt = os.path.join(out_dir, "q_encoder.pt")
    torch.save(sd, ckpt)

    # ---- questions ------------------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(seed)
    qas = os.path.join(out_dir, "qas.json")
    with open(qas, "w") as f:
        for i in range(n_q):
            n = int(rng.integers(8, 41))
            a, b = (int(x) for x in rng.integers(0, rows, 2))
            f.write(json.dumps({"_id": f"q{i}", "question": " ".join(words[j] for j in rng.integers(0, W, n)) + "?", "answer": ["the"],
                                "sp": [f"T{a}", f"T{b if b != a else (a + 1) % rows}"], "type": "bridge" if i % 5 else "comparison"}) + "\n")
    log(f"[assets] done {time.time() - t0:.1f}s")
    return {"raw_data": qas, "indexpath": index_path, "corpus_dict": store, "model_path": ckpt, "model_name": model_dir, "build_seconds": round(time.time() - t0, 1)}

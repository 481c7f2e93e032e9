"""oracle/gen_golden.py -- TEST INFRASTRUCTURE: regenerates tests/golden/*.

Run in the BUILD container only (it imports the Python reference from /root/reference, which never
travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

What is pinned by the REFERENCE ITSELF (imported, executed here):
  encoder_*.npz   RobertaRetriever.encode_q  (mdr/retrieval/models/mhop_retriever.py:12-41) and
                  RobertaCtxEncoder.forward  (mdr/retrieval/models/retriever.py:176-190) outputs, with
                  weights derived from oracle/seeded.py (seed, name, shape) and loaded through the
                  reference's own load_saved (mdr/retrieval/utils/utils.py:10-22).
  load_saved.npz  load_saved behaviour: `module.` prefix, extra keys tolerated, missing key raises.
  collate.npz     collate_tokens (mdr/retrieval/data/data_utils.py:11-29) and em_collate
                  (mdr/retrieval/data/encode_datasets.py:102-114) on ragged inputs.
What is pinned by float64 ground truth (FAISS is not in /root/reference and not installed):
  mips_*.npz      D,I = top-k of x @ xb.T computed in float64 numpy, stable (score desc, id asc).
What is a frozen restatement (the code is inline under __main__ and cannot be imported):
  mhop.json       beam aggregation, metrics, JSONL and log lines from oracle/mhop_oracle.py.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import mhop_oracle, roberta_oracle, seeded  # noqa: E402


def _ref_models(geom, sd_np, tmp):
    import torch
    from transformers import RobertaConfig, RobertaModel
    from mdr.retrieval.models.mhop_retriever import RobertaRetriever
    from mdr.retrieval.models.retriever import RobertaCtxEncoder
    from mdr.retrieval.utils.utils import load_saved

    cfg = RobertaConfig(vocab_size=geom["vocab"], hidden_size=geom["hidden"], num_hidden_layers=geom["layers"],
                        num_attention_heads=geom["heads"], intermediate_size=geom["ffn"],
                        max_position_embeddings=geom["max_pos"], type_vocab_size=1, layer_norm_eps=geom["ln_eps"],
                        pad_token_id=1, bos_token_id=0, eos_token_id=2, hidden_act="gelu",
                        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    mdir = os.path.join(tmp, f"hf_{geom['hidden']}")
    RobertaModel(cfg).save_pretrained(mdir)
    args = types.SimpleNamespace(model_name=mdir)
    ckpt = os.path.join(tmp, f"ckpt_{geom['hidden']}.pt")
    torch.save({("module." + k): torch.from_numpy(v) for k, v in sd_np.items()}, ckpt)
    q = load_saved(RobertaRetriever(cfg, args), ckpt, exact=False).eval()
    c = load_saved(RobertaCtxEncoder(cfg, args), ckpt, exact=False).eval()
    for m in (q, c):
        try:
            m.encoder.set_attn_implementation("eager")
        except Exception as e:  # pragma: no cover
            print("warn: could not force eager attention:", e)
    return q, c, cfg, args, ckpt


def o1_mode(literal):
    """apex-O1 numerics (the regime the reference publishes its numbers under, eval_mhop_retrieval.py:86-90) imposed on the
    IMPORTED reference model on the CPU, as a torch-function mode: every Linear / matmul ("whitelist" ops, run by apex in fp16 on
    tensor cores with fp32 accumulation) sees its inputs and weights ROUNDED TO FP16; LayerNorm, softmax, GELU ("blacklist" ops)
    and the embedding sums stay fp32.
      literal = False ("o1ops"): only the operand rounding -- products and sums in fp32, outputs not rounded. This is the dataflow of
                the HIP encoder in residual_fp32 mode, so HIP-vs-this isolates kernel error from the regime's own rounding.
      literal = True  ("o1lit"): additionally what apex O1 literally does with the OUTPUTS of those ops: a Linear returns an fp16
                tensor (matmul result rounded to fp16, then `+= bias` in fp16: torch's F.linear for 3-D inputs), torch.matmul
                returns fp16 (attention scores and context). More roundings than the HIP path performs."""
    import torch
    import torch.nn.functional as F
    from torch.overrides import TorchFunctionMode

    def r(t):
        return t.half().float()

    class Mode(TorchFunctionMode):
        def __torch_function__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            if func is F.linear:
                x, w = args[0], args[1]
                b = args[2] if len(args) > 2 else kwargs.get("bias")
                y = func(r(x), r(w))
                if literal:
                    y = r(y)
                    return r(y + r(b)) if b is not None else y
                return y + b if b is not None else y
            if func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.bmm):
                y = func(r(args[0]), r(args[1]))
                return r(y) if literal else y
            return func(*args, **kwargs)

    return Mode()


def gen_encoder(tmp):
    import torch
    cases = {
        "tiny": (seeded.TINY, 11, [("q", 4, 70, 1), ("qsp", 3, 350, 1), ("ctx", 5, 97, 0), ("one", 1, 8, 1)]),
        "base": (seeded.ROBERTA_BASE, 7, [("q", 3, 70, 1), ("qsp", 2, 350, 1), ("ctx", 2, 300, 0)]),
        "wide2": (seeded.WIDE2, 13, [("q", 6, 70, 1), ("qsp", 3, 350, 1)]),
    }
    for tag, (geom, seed, batches) in cases.items():
        sd = seeded.make_state_dict(seed, geom)
        q, c, *_ = _ref_models(geom, sd, tmp)
        out = {"seed": np.int64(seed)}
        for name, B, L, pad_fill in batches:
            ids, mask = seeded.make_token_batch(seed, f"{tag}.{name}", B, L, geom["vocab"], pad_fill=pad_fill)
            with torch.no_grad():
                eq = q.encode_q(torch.from_numpy(ids), torch.from_numpy(mask), None).numpy()
                ec = c({"input_ids": torch.from_numpy(ids), "input_mask": torch.from_numpy(mask)})["embed"].numpy()
            assert np.array_equal(eq, ec), "RobertaRetriever.encode_q and RobertaCtxEncoder.forward differ"
            mine = roberta_oracle.encode(sd, geom, ids, mask, np.float64)
            print(f"encoder {tag}.{name}: ref-vs-restatement(f64) max abs diff {np.abs(mine - eq).max():.3e}, "
                  f"|out| mean {np.abs(eq).mean():.3f}")
            out[f"{name}.ids"], out[f"{name}.mask"], out[f"{name}.embed"] = ids, mask, eq.astype(np.float32)
            # the same reference model under apex-O1 numerics (operand rounding only / literal): VERDICT r2 item 2
            for key, literal in (("embed_o1ops", False), ("embed_o1lit", True)):
                with torch.no_grad(), o1_mode(literal):
                    eo = q.encode_q(torch.from_numpy(ids), torch.from_numpy(mask), None).numpy()
                from oracle import roberta_torch
                mine_o = roberta_torch.encode(sd, geom, ids, mask, torch.float32, "cpu", o1="literal" if literal else "operands").numpy()
                print(f"    {key}: |o1 - fp32| max {np.abs(eo - eq).max():.3e} mean {np.abs(eo - eq).mean():.3e}; "
                      f"restatement(o1) vs reference(o1) max {np.abs(mine_o - eo).max():.3e}")
                out[f"{name}.{key}"] = eo.astype(np.float32)
        np.savez_compressed(os.path.join(GOLD, f"encoder_{tag}.npz"), **out)


def gen_load_saved(tmp):
    import torch
    geom, seed = seeded.TINY, 23
    sd = seeded.make_state_dict(seed, geom)
    q, _, cfg, args, _ = _ref_models(geom, sd, tmp)
    from mdr.retrieval.models.mhop_retriever import RobertaRetriever
    from mdr.retrieval.utils.utils import load_saved
    ids, mask = seeded.make_token_batch(seed, "ls", 2, 16, geom["vocab"])
    res = {"seed": np.int64(seed), "ids": ids, "mask": mask}

    def run(tag, state):
        p = os.path.join(tmp, f"ls_{tag}.pt")
        torch.save(state, p)
        try:
            m = load_saved(RobertaRetriever(cfg, args), p, exact=False).eval()
            m.encoder.set_attn_implementation("eager")
            with torch.no_grad():
                res[f"{tag}.embed"] = m.encode_q(torch.from_numpy(ids), torch.from_numpy(mask), None).numpy()
            res[f"{tag}.raises"] = np.bool_(False)
        except Exception as e:
            print(f"load_saved[{tag}] raised {type(e).__name__}")
            res[f"{tag}.raises"] = np.bool_(True)

    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    run("plain", dict(t))
    run("module_prefix", {"module." + k: v for k, v in t.items()})
    run("extra_keys", {**t, "encoder.embeddings.position_ids": torch.arange(514)[None], "foo.bar": torch.zeros(3)})
    miss = dict(t)
    del miss["project.0.bias"]
    run("missing_key", miss)
    nopool = {k: v for k, v in t.items() if "pooler" not in k}
    run("no_pooler", nopool)  # the reference model HAS a pooler -> strict load raises
    np.savez_compressed(os.path.join(GOLD, "load_saved.npz"), **res)


def gen_collate():
    import torch
    from mdr.retrieval.data.data_utils import collate_tokens
    from mdr.retrieval.data.encode_datasets import em_collate
    lens = [5, 1, 9, 3]
    vals = [torch.arange(10, 10 + n, dtype=torch.long) for n in lens]
    out = {"lens": np.array(lens)}
    out["pad0"] = collate_tokens(vals, 0).numpy()
    out["pad1"] = collate_tokens(vals, 1).numpy()
    out["left"] = collate_tokens(vals, 1, left_pad=True).numpy()
    samples = [{"input_ids": v.view(1, -1), "attention_mask": torch.ones(1, len(v), dtype=torch.long)} for v in vals]
    b = em_collate(samples)
    out["em.input_ids"], out["em.input_mask"] = b["input_ids"].numpy(), b["input_mask"].numpy()
    assert em_collate([]) == {}
    np.savez_compressed(os.path.join(GOLD, "collate.npz"), **out)


def mips_truth(x, xb, k):
    S = x.astype(np.float64) @ xb.astype(np.float64).T
    nb = xb.shape[0]
    kk = min(k, nb)
    order = np.argsort(-S, axis=1, kind="stable")[:, :kk]  # score desc, id asc
    D = np.full((x.shape[0], k), -np.finfo(np.float32).max, np.float64)
    I = np.full((x.shape[0], k), -1, np.int64)
    D[:, :kk] = np.take_along_axis(S, order, 1)
    I[:, :kk] = order
    # gap to the first excluded score (inf if none): tests compare ids only where gap > tolerance
    if kk < nb:
        nxt = -np.sort(-S, axis=1)[:, kk]
        gap = D[:, kk - 1] - nxt
    else:
        gap = np.full(x.shape[0], np.inf)
    return D, I, gap


def gen_mips():
    d = 768
    xb = seeded.normal(0, "mips.xb", (4096, d))
    # planted duplicates: rows 100 and 3000 equal row 7; rows 4000..4003 equal row 50 (pins the tie rule)
    xb[100] = xb[7]
    xb[3000] = xb[7]
    xb[4000:4004] = xb[50]
    x = seeded.normal(1, "mips.x", (37, d))
    x[0] = xb[7] * 1.0          # exact-tie query: best three are ids 7,100,3000 in that order
    x[1] = xb[50] * 0.5
    x[2] = xb[9] + 0.05 * seeded.normal(2, "mips.noise", (d,))  # planted near-neighbour
    out = {"xb_seed": np.int64(0), "x": x}
    out["dup_rows"] = np.array([[100, 7], [3000, 7], [4000, 50], [4001, 50], [4002, 50], [4003, 50]])
    for nq in (5, 37):
        for k in (1, 4, 8, 100):
            D, I, gap = mips_truth(x[:nq], xb, k)
            out[f"nq{nq}.k{k}.D"], out[f"nq{nq}.k{k}.I"], out[f"nq{nq}.k{k}.gap"] = D, I, gap
    # fewer rows than k: -FLT_MAX / -1 padding (FAISS behaviour)
    D, I, gap = mips_truth(x[:5], xb[:6], 8)
    out["short.D"], out["short.I"] = D, I
    np.savez_compressed(os.path.join(GOLD, "mips_4096x768.npz"), **out)
    # d != 768 and ragged N (not a multiple of any tile)
    xb2 = seeded.normal(3, "mips.xb2", (1000 + 37, 128))
    x2 = seeded.normal(4, "mips.x2", (21, 128))
    out2 = {"x": x2}
    for k in (1, 5, 64):
        D, I, gap = mips_truth(x2, xb2, k)
        out2[f"k{k}.D"], out2[f"k{k}.I"], out2[f"k{k}.gap"] = D, I, gap
    np.savez_compressed(os.path.join(GOLD, "mips_1037x128.npz"), **out2)


def gen_mhop():
    id2doc_list = {"0": ["Alpha", "alpha text", False], "1": ["Beta", "beta text", True], "2": ["Gamma", "  ", False],
                   "3": ["Delta", "delta text", False], "4": ["Beta", "other beta", False], "5": ["Eps", "eps text", False]}
    id2doc = mhop_oracle.normalise_id2doc(dict(id2doc_list))
    items = [
        {"_id": "q0", "question": "Who is alpha and beta?", "answer": ["x"], "sp": ["Alpha", "Beta"], "type": "bridge"},
        {"_id": "q1", "question": "gamma or delta??", "answer": ["yes"], "sp": ["Gamma", "Delta"], "type": "comparison"},
        {"_id": "q2", "question": "no question mark", "answer": ["z"], "sp": ["Eps", "Alpha"], "type": "bridge"},
    ]
    rng = np.random.default_rng(5)
    cases = []
    for beam, topk in [(1, 1), (2, 2), (2, 4), (3, 2)]:
        B = len(items)
        D = rng.standard_normal((B, beam)).astype(np.float32)
        D = -np.sort(-D, axis=1)
        I = np.stack([rng.permutation(6)[:beam] for _ in range(B)]).astype(np.int64)
        if beam > 1:
            I[1, 1] = 2  # the empty-text doc in a beam slot -> -inf hop-1 score
        D2 = -np.sort(-rng.standard_normal((B * beam, beam)).astype(np.float32), axis=1)
        I2 = np.stack([rng.permutation(6)[:beam] for _ in range(B * beam)]).astype(np.int64)
        qs = [mhop_oracle.strip_question(it["question"]) for it in items]
        Dw = D.copy()
        pairs = mhop_oracle.build_hop2_pairs(qs, Dw, I, id2doc)
        chains = mhop_oracle.rank_paths(Dw, I, D2, I2, beam, topk)
        metrics, records = [], []
        for it, ch in zip(items, chains):
            m = mhop_oracle.question_metrics(ch, it["sp"], id2doc)
            m.update(question=it["question"], type=it["type"])
            metrics.append(m)
            records.append(json.dumps(mhop_oracle.output_record(it, ch, id2doc)))
        cases.append({
            "beam": beam, "topk": topk, "D": D.tolist(), "I": I.tolist(), "D2": D2.tolist(), "I2": I2.tolist(),
            "stripped": qs, "pairs": pairs,
            "D_after": [[None if np.isinf(v) else float(v) for v in row] for row in Dw],
            "chains": [[[h1, h2, (None if np.isinf(s) else s)] for h1, h2, s in ch] for ch in chains],
            "metrics": metrics, "jsonl": records, "log": mhop_oracle.summary_lines(metrics),
        })
    with open(os.path.join(GOLD, "mhop.json"), "w") as f:
        json.dump({"id2doc_list": id2doc_list, "items": items, "cases": cases}, f, indent=1)


def gen_answer_recall():
    """Outputs of the reference's own para_has_answer / SimpleTokenizer (mdr/retrieval/utils/utils.py:126-139,
    basic_tokenizer.py:238-277) and of the literal concatenation + log expressions of eval_mhop_retrieval.py:208-217,269-273
    (inline under __main__, restated here) on hand-made paragraphs: unicode forms, punctuation, case, token boundaries."""
    from mdr.retrieval.utils.basic_tokenizer import SimpleTokenizer
    from mdr.retrieval.utils.utils import para_has_answer
    tok = SimpleTokenizer()
    paras = [
        "Barack Obama was born in Honolulu, Hawaii (U.S.A.) on August 4, 1961.",
        "The caf\u00e9 Fran\u00e7ois opened in 1999; it's \u201cfamous\u201d for cr\u00e8me br\u00fbl\u00e9e.",
        "cafe\u0301 with a combining accent; na\u00efve co-operate re\u2011enter state-of-the-art 3.14 1,000 $5 #tag @me",
        "\u6771\u4eac\u90fd is Tokyo.\tTabs\nand newlines\u00a0and nbsp \u200b zero width. \u0130stanbul STRASSE stra\u00dfe",
        "",
        "yes no Alpha alpha textBeta beta text",
    ]
    answers = [["Honolulu"], ["honolulu , hawaii"], ["Hawaii U.S.A"], ["u . s . a ."], ["August 4 1961"], ["August 4, 1961"], ["4,"],
               ["Cafe Francois"], ["caf\u00e9 fran\u00e7ois"], ["cafe\u0301"], ["caf\u00e9"], ["it s"], ["it's"], ["\u201cfamous\u201d"], ["famous"],
               ["creme brulee"], ["cr\u00e8me br\u00fbl\u00e9e"], ["naive"], ["na\u00efve"], ["co operate"], ["co-operate"], ["re enter"],
               ["state of the art"], ["3.14"], ["3 . 14"], ["1,000"], ["1000"], ["$ 5"], ["# tag"], ["\u6771\u4eac\u90fd"], ["\u6771\u4eac"],
               ["Tokyo Tabs"], ["tabs and"], ["and nbsp zero"], ["nbsp zero width"], ["istanbul"], ["\u0130stanbul"], ["strasse"],
               ["stra\u00dfe"], ["STRASSE stra\u00dfe"], [""], ["   "], ["yes"], ["textbeta"], ["text beta"], ["alpha text", "nothing"],
               ["nothing", "beta text"], ["nothing at all"]]
    tokens = [tok.tokenize(__import__("unicodedata").normalize("NFD", p)).words(uncased=True) for p in paras]
    table = [[bool(para_has_answer(a, p, tok)) for a in answers] for p in paras]
    # the inline CLI expressions (eval_mhop_retrieval.py:209-217 and :269-273)
    id2doc = {"0": {"title": "Alpha", "text": "alpha text"}, "1": {"title": "Beta", "text": "beta text"},
              "2": {"title": "Gamma", "text": ""}, "3": {"title": "Delta", "text": "delta \u00e9t\u00e9 1999"}}
    items = [{"question": "q0?", "answer": ["beta text"], "type": "bridge"}, {"question": "q1", "answer": ["ete 1999", "\u00e9t\u00e9 1999"]},
             {"question": "q2", "answer": ["textBeta"], "type": "bridge"}, {"question": "q3", "answer": ["missing"], "type": "comparison"}]
    chains = [[("0", "1")], [("2", "3"), ("0", "0")], [("0", "1"), ("1", "0")], [("3", "2")]]
    concat, metrics = [], []
    for it, paths in zip(items, chains):
        concat_p = "yes no "
        for p in paths:
            concat_p += " ".join([id2doc[doc_id]["title"] + " " + id2doc[doc_id]["text"] for doc_id in p])
        concat.append(concat_p)
        metrics.append({"question": it["question"], "ans_recall": int(para_has_answer(it["answer"], concat_p, tok)), "type": it.get("type", "single")})
    import collections
    type2items = collections.defaultdict(list)
    for m in metrics:
        type2items[m["type"]].append(m)
    log = [f"Evaluating {len(metrics)} samples...", f'Ans Recall: {np.mean([m["ans_recall"] for m in metrics])}']
    for t in type2items.keys():
        log.append(f"{t} Questions num: {len(type2items[t])}")
        log.append(f'Ans Recall: {np.mean([m["ans_recall"] for m in type2items[t]])}')
    with open(os.path.join(GOLD, "answer_recall.json"), "w") as f:
        json.dump({"paras": paras, "answers": answers, "tokens": tokens, "has_answer": table, "id2doc": id2doc, "items": items,
                   "chains": [[list(p) for p in c] for c in chains], "concat": concat, "metrics": metrics, "log": log}, f, indent=1)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    only = set(sys.argv[1:])
    with tempfile.TemporaryDirectory() as tmp:
        if not only or "mips" in only:
            gen_mips()
        if not only or "mhop" in only:
            gen_mhop()
        if not only or "collate" in only:
            gen_collate()
        if not only or "answer_recall" in only:
            gen_answer_recall()
        if not only or "load_saved" in only:
            gen_load_saved(tmp)
        if not only or "encoder" in only:
            gen_encoder(tmp)
    print("golden files:", sorted(os.listdir(GOLD)))

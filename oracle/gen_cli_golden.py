"""oracle/gen_cli_golden.py -- TEST INFRASTRUCTURE: regenerates tests/golden/cli_ref.{json,npz}.

Run in the BUILD container only (it executes the Python reference from /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_cli_golden.py

What this pins, and by what.  The two-hop batch loop of the reference is inline under `__main__` in
/root/reference/scripts/eval/eval_mhop_retrieval.py:50-284 and cannot be imported, so rounds 1-4 checked the product's host logic
(mhop.build_hop2_pairs / rank_paths / question_metrics / output_record / summary_lines) against the builder's own restatement
(oracle/mhop_oracle.py).  Here the reference's script ITSELF is executed -- `runpy.run_path(<that file>, run_name="__main__")`, every line of
:50-284 that runs is the reference's -- on toy assets, and what it computed is captured:
  * per batch the (D, I) and (D_, I_) that `index.search` returned and the embeddings it was asked with,
  * the (question, passage) pairs it handed to the tokenizer for hop 2 (:158-166, incl. the empty-text -> title / -inf rule),
  * the log lines it wrote (:71-284) and the bytes of the --save-path JSONL (:260-263), its `metrics` list (:211-242).

The STUBS stand in for LIBRARIES the image lacks, nothing else:
  faiss        `IndexFlatIP(d)`, `.add`, `.search` -> exact inner products in float32 numpy, best first, ties by ascending id (the rule of
               oracle/flat_ip_oracle.c); `StandardGpuResources` / `index_cpu_to_gpu` -> identity (there is no GPU here).
  apex.amp     `initialize(model, opt_level='O1')` -> the model unchanged (fp32 on the CPU).
  cuda         `model.to(torch.device('cuda'))` and `move_to_cuda` -> no-ops (CPU tensors).
  tqdm         identity iterator (keeps the captured stderr to the log lines).
  transformers `AutoTokenizer.from_pretrained` returns an adapter with transformers 2.11's `batch_encode_plus(x, max_length=n,
               pad_to_max_length=True, return_tensors="pt")` on top of the installed tokenizer's BPE: per segment the 2.11 prefix space, the
               `<s> A </s></s> B </s>` template, the literal 2.11 `truncate_sequences('longest_first')` pop loop, right padding.  (2.11 itself
               is not installable offline: the token ids stay UNPINNED; the host loop does not depend on them.)
The model is the reference's own `RobertaRetriever` (mdr/retrieval/models/mhop_retriever.py:12-41) loaded by its own `load_saved`
(mdr/retrieval/utils/utils.py:10-22) from a checkpoint derived from oracle/seeded.py, so the GPU test can rebuild the very same assets on the GPU
box from (seed, name, shape) + tests/golden/tiny_bpe and compare the drop-in CLI's chains with the captured ones.

The corpus encoder (scripts/encode_corpus.py) is executed the same way (`run_reference_encode_corpus`: its EmDataset, em_collate,
RobertaCtxEncoder and np.save on the toy corpus; the `id2doc.json` it wrote and a sample of the rows of its `.npy` are kept).
The FEVER variant (scripts/eval/eval_mhop_fever.py) is executed the same way (`run_reference_fever`: its own code object, the same stubs;
its final write to a directory on its author's machine fails here as it does everywhere else, after `retrieval_outputs` has been filled).

Assets (`build_assets`, shared with the tests -- it needs only numpy / torch / transformers, not the reference): 257 passages, four with empty
text, three with identical embeddings (exact path-score ties), 23 questions (one ending in `??`, one without `?`, five yes/no answers), a list-valued
and a dict-valued corpus dict, a 2-layer 768-wide checkpoint with the `module.` prefix, corpus embeddings = seeded normals.
"""
import contextlib
import hashlib
import io
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
SCRIPT = os.path.join(REF, "scripts", "eval", "eval_mhop_retrieval.py")
sys.dont_write_bytecode = True
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import seeded  # noqa: E402

N_DOCS, N_Q, SEED = 257, 23, 41
MAX_Q_LEN, MAX_Q_SP_LEN, BATCH = 12, 40, 10
EMPTY_DOCS = (5, 77, 130, 200)
DUPLICATE_ROWS = ((100, 73), (201, 175), (150, 107), (151, 213))  # xb[100] = xb[73], xb[201] = xb[175] (rows a few questions retrieve): equal hop scores -> equal path scores
# supporting titles by question number mod 4: chosen among the passages the toy encoder retrieves most, so that every metric takes both values
SP_BY_RESIDUE = (["T173", "T58"], ["T173", "Zürich"], ["T58", "T-absent"], None)  # "Zürich" = passage 213's title (precomposed u-umlaut)
# round 6: strings no fixture carried before. Titles with precomposed / combining / padded characters (EmDataset NFD-normalises and strips the title
# it tokenises, encode_datasets.py:95, and writes the RAW title to id2doc.json, :76-78); passage texts holding the tokenizer's own special-token strings; a question that
# ends in " ?" (one "?" is stripped, :139: the trailing blank stays) and one with leading blanks.
TITLE_OVERRIDES = {213: "Zürich", 214: "Krako\u0301w", 215: "  Kraków  ", 216: "Ångström (unit)"}
TEXT_SUFFIXES = {58: " <s> inner </s> <mask> tail", 173: " so <mask> it </s>", 12: " <s>"}
Q_TRAILING_BLANK, Q_LEADING_BLANKS = 5, 6
# (beam, topk, corpus-dict shape, extra flags).  (50, 50): the reference's own downstream setting (README.md:240-241 b50_k50), JSONL kept as a hash.
# (100, 100): README.md:241 b100_k100, on the first N_Q_SMALL questions only (a 100 x 100 beam grid per question: the capture stays small).
# "base12" (round 6): the same run with a 12-layer, ffn-3072 checkpoint (seeded.ROBERTA_BASE geometry on the toy vocabulary) -- the real depth.
CASES = [(1, 1, "list", []), (3, 4, "list", []), (5, 2, "dict", []), (3, 4, "dict", ["--only-eval-ans"]), (50, 50, "list", []), (100, 100, "dict", ["small"]),
         (1, 1, "list", ["base12"])]
N_Q_SMALL = 5


def tiny_tokenizer():
    import transformers
    bpe = os.path.join(GOLD, "tiny_bpe")
    vocab = json.load(open(os.path.join(bpe, "vocab.json")))
    merges = [tuple(ln.split()) for ln in open(os.path.join(bpe, "merges.txt")).read().split("\n") if ln and not ln.startswith("#")]
    return transformers.RobertaTokenizer(vocab=vocab, merges=merges)


def build_assets(out_dir):
    """Writes the toy assets under out_dir and returns their paths + the in-memory pieces.  Deterministic: seeded.py + fixed word lists."""
    import torch
    import transformers
    os.makedirs(out_dir, exist_ok=True)
    tok = tiny_tokenizer()
    geom = dict(seeded.TINY, hidden=768, heads=12, ffn=512, vocab=max(seeded.TINY["vocab"], len(tok)))  # the script hard-codes d = 768 (:93)
    sd = seeded.make_state_dict(SEED, geom)
    model_dir = os.path.join(out_dir, "toy-roberta")  # "roberta" in --model-name switches the empty-passage rule on (:161)
    cfg = transformers.RobertaConfig(vocab_size=geom["vocab"], hidden_size=768, num_hidden_layers=geom["layers"], num_attention_heads=12,
                                     intermediate_size=512, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1,
                                     bos_token_id=0, eos_token_id=2, hidden_act="gelu")
    cfg.save_pretrained(model_dir)
    tok.save_pretrained(model_dir)
    ckpt = os.path.join(out_dir, "q_encoder.pt")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)

    words = ("the quick brown fox jumps over lazy dog river bank retrieval encoder index beam passage question answer Paris London film band "
             "album studio producer capital France Seine stadium people author born city population Zürich Kraków 1950 2012 80,000 3.14").split()
    pick = seeded.integers(SEED, "cli.words", (N_DOCS + N_Q, 48), 0, len(words))
    lens = seeded.integers(SEED, "cli.lens", (N_DOCS + N_Q,), 4, 48)
    docs = [{"title": f"T{i}", "text": " ".join(words[j] for j in pick[i, :lens[i]])} for i in range(N_DOCS)]
    for i in EMPTY_DOCS:
        docs[i]["text"] = "" if i % 2 else "  \t"
    docs[9]["title"] = "T8"  # two passages under one title: the metrics work on titles (:219-242)
    for i, t in TITLE_OVERRIDES.items():
        docs[i]["title"] = t
    for i, suf in TEXT_SUFFIXES.items():
        docs[i]["text"] += suf
    xb = seeded.normal(SEED, "cli.xb", (N_DOCS, 768))
    for dst, src in DUPLICATE_ROWS:
        xb[dst] = xb[src]
    index_path = os.path.join(out_dir, "index.npy")
    np.save(index_path, xb)
    id2doc_list = {str(i): [d["title"], d["text"], bool(i % 3 == 0)] for i, d in enumerate(docs)}
    id2doc_dict = {str(i): {"title": d["title"], "text": d["text"]} for i, d in enumerate(docs)}
    paths = {"list": os.path.join(out_dir, "id2doc_list.json"), "dict": os.path.join(out_dir, "id2doc_dict.json")}
    json.dump(id2doc_list, open(paths["list"], "w"))
    json.dump(id2doc_dict, open(paths["dict"], "w"))
    qs = []
    for i in range(N_Q):
        r = N_DOCS + i
        text = " ".join(words[j] for j in pick[r, :4 + lens[r] % 9])
        q = text + ("??" if i == 3 else "" if i == 4 else " ?" if i == Q_TRAILING_BLANK else "?")
        if i == Q_LEADING_BLANKS:
            q = "  " + q
        ans = ["yes"] if i % 5 == 0 else ["no"] if i == 7 else [(docs[(i * 7) % N_DOCS]["text"].split() or ["Seine"])[0] if i % 2 else "zzz-not-there", "not in any passage"]
        qs.append({"_id": f"q{i}", "question": q, "answer": ans, "sp": SP_BY_RESIDUE[i % 4] or [f"T{(i * 11) % N_DOCS}", f"T{(i * 11 + 1) % N_DOCS}"],
                   "type": "bridge" if i % 3 else "comparison"})
    raw = os.path.join(out_dir, "qas.json")
    with open(raw, "w") as f:
        f.write("\n".join(json.dumps(q) for q in qs))
    corpus_jsonl = os.path.join(out_dir, "corpus.jsonl")
    with open(corpus_jsonl, "w") as f:
        f.write("\n".join(json.dumps(dict(d, intro=True) if i % 3 == 0 else d) for i, d in enumerate(docs)))
    # the other corpus formats EmDataset reads (encode_datasets.py:56-72): a TSV with the `id<TAB>text<TAB>title` header row, and a JSONL whose PATH contains "fever"
    import csv
    corpus_tsv = os.path.join(out_dir, "corpus.tsv")
    with open(corpus_tsv, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t")
        w.writerow(["id", "text", "title"])
        for i, d in enumerate(docs):
            w.writerow([str(i), d["text"], d["title"]])
    corpus_fever = os.path.join(out_dir, "fever_corpus.jsonl")
    with open(corpus_fever, "w") as f:
        f.write(open(corpus_jsonl).read())
    # --is_query_embed (encode_datasets.py:52-54,82): a JSONL of records that still need "title" and "text" (its __getitem__ reads both), cut at --max_q_len, no id2doc.json
    queries_embed = os.path.join(out_dir, "queries_embed.jsonl")
    with open(queries_embed, "w") as f:
        f.write("\n".join(json.dumps({"title": d["title"], "text": d["text"], "question": q["question"]}) for d, q in zip(docs[200:200 + N_Q], qs)))
    claims = [{"id": 1000 + i, "claim": q["question"].rstrip("?"), "label": "SUPPORTS" if i % 2 else "REFUTES"} for i, q in enumerate(qs)]
    raw_fever = os.path.join(out_dir, "claims.json")
    with open(raw_fever, "w") as f:
        f.write("\n".join(json.dumps(c) for c in claims))
    raw_small = os.path.join(out_dir, "qas_small.json")
    with open(raw_small, "w") as f:
        f.write("\n".join(json.dumps(q) for q in qs[:N_Q_SMALL]))
    return {"tok": tok, "geom": geom, "sd": sd, "model_dir": model_dir, "ckpt": ckpt, "index": index_path, "xb": xb, "id2doc": paths, "raw": raw, "raw_small": raw_small, "raw_fever": raw_fever, "claims": claims, "corpus_jsonl": corpus_jsonl,
            "corpus_tsv": corpus_tsv, "corpus_fever": corpus_fever, "queries_embed": queries_embed, "questions": qs, "docs": docs}


def build_base12_assets(a):
    """The 12-layer checkpoint + model directory of the "base12" case (340 MB: built only by the generator and by the GPU test that needs it)."""
    import torch
    import transformers
    if "ckpt_base12" in a:
        return a
    out_dir = os.path.dirname(a["ckpt"])
    geom = dict(seeded.ROBERTA_BASE, vocab=a["geom"]["vocab"])
    sd = seeded.make_state_dict(SEED + 1, geom)
    model_dir = os.path.join(out_dir, "toy-roberta-base12")
    cfg = transformers.RobertaConfig(vocab_size=geom["vocab"], hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                     max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                                     hidden_act="gelu")
    cfg.save_pretrained(model_dir)
    a["tok"].save_pretrained(model_dir)
    ckpt = os.path.join(out_dir, "q_encoder_base12.pt")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    a.update(ckpt_base12=ckpt, model_dir_base12=model_dir, geom_base12=geom)
    return a


def cli_argv(a, beam, topk, shape, extra, save):
    small = "small" in extra  # (not a flag of the script: selects the short question file)
    base12 = "base12" in extra  # (not a flag either: selects the 12-layer checkpoint)
    extra = [e for e in extra if e not in ("small", "base12")]
    ckpt, model_dir = (a["ckpt_base12"], a["model_dir_base12"]) if base12 else (a["ckpt"], a["model_dir"])
    return [a["raw_small"] if small else a["raw"], a["index"], a["id2doc"][shape], ckpt, "--batch-size", str(BATCH), "--beam-size", str(beam), "--topk", str(topk),
            "--model-name", model_dir, "--gpu", "--shared-encoder", "--save-path", save, "--max-q-len", str(MAX_Q_LEN),
            "--max-q-sp-len", str(MAX_Q_SP_LEN)] + list(extra)


# --------------------------------------------------------------------------------------------------------------------------------------
# library stubs
# --------------------------------------------------------------------------------------------------------------------------------------
class Capture:
    def __init__(self):
        self.searches, self.tokenizer_calls, self.encode_plus_calls = [], [], []


def faiss_stub(cap):
    m = types.ModuleType("faiss")

    class IndexFlatIP:
        def __init__(self, d):
            self.d, self.xb = d, np.zeros((0, d), np.float32)

        def add(self, x):
            assert x.dtype == np.float32 and x.shape[1] == self.d
            self.xb = np.concatenate([self.xb, x])

        def search(self, x, k):
            assert x.dtype == np.float32 and x.flags["C_CONTIGUOUS"]
            s = x @ self.xb.T
            I = np.argsort(-s, axis=1, kind="stable")[:, :k].astype(np.int64)  # best first, ties by ascending id
            D = np.take_along_axis(s, I, axis=1).astype(np.float32)
            cap.searches.append({"x": x.copy(), "D": D.copy(), "I": I.copy()})  # copies: the script writes -inf into D afterwards (:165)
            return D, I

    m.IndexFlatIP = IndexFlatIP
    m.StandardGpuResources = lambda: None
    m.index_cpu_to_gpu = lambda res, dev, index: index
    m.write_index = lambda *a: (_ for _ in ()).throw(RuntimeError("--save-index is not part of the toy runs"))
    return m


class RobertaTokenizer211:
    """transformers 2.11's `batch_encode_plus` / `encode_plus` contract for a RoBERTa tokenizer, on the installed tokenizer's BPE (see the header). (The class
    name carries "Roberta": EmDataset's empty-text rule looks at `tokenizer.__class__.__name__`, encode_datasets.py:88.)"""

    def __init__(self, tok, cap):
        self.tok, self.cap = tok, cap

    def _bpe(self, text):
        if text and not text[0].isspace():  # RobertaTokenizer.prepare_for_tokenization, add_prefix_space defaulting to add_special_tokens
            text = " " + text
        return list(self.tok(text, add_special_tokens=False, truncation=False)["input_ids"])

    def encode_plus(self, text, text_pair=None, max_length=None, return_tensors=None):
        """`encode_plus(title, text_pair=text, max_length=n, return_tensors="pt")` (encode_datasets.py:95): special tokens, longest-first truncation, NO padding."""
        import torch
        assert return_tensors == "pt" and max_length
        self.cap.encode_plus_calls.append([text, text_pair, max_length])  # what the reference's EmDataset hands to its tokenizer (encode_datasets.py:95)
        bos, eos = self.tok.bos_token_id, self.tok.eos_token_id
        ia, ib = self._bpe(text), (self._bpe(text_pair) if text_pair is not None else None)
        over = len(ia) + (len(ib) if ib is not None else 0) + (4 if ib is not None else 2) - max_length
        for _ in range(max(over, 0)):
            if ib is None or len(ia) > len(ib):
                ia = ia[:-1]
            else:
                ib = ib[:-1]
        row = [bos] + ia + [eos] + ([eos] + ib + [eos] if ib is not None else [])
        return {"input_ids": torch.tensor([row], dtype=torch.long), "attention_mask": torch.ones((1, len(row)), dtype=torch.long)}

    def batch_encode_plus(self, batch, max_length=None, pad_to_max_length=False, return_tensors=None):
        import torch
        assert pad_to_max_length and return_tensors == "pt" and max_length
        self.cap.tokenizer_calls.append([list(x) if isinstance(x, tuple) else x for x in batch])
        bos, eos, pad = self.tok.bos_token_id, self.tok.eos_token_id, self.tok.pad_token_id
        ids_rows, mask_rows = [], []
        for item in batch:
            a, b = (item if isinstance(item, (tuple, list)) else (item, None))
            ia, ib = self._bpe(a), (self._bpe(b) if b is not None else None)
            over = len(ia) + (len(ib) if ib is not None else 0) + (4 if ib is not None else 2) - max_length
            for _ in range(max(over, 0)):  # truncate_sequences(..., 'longest_first'): one token at a time, the second sequence on a tie
                if ib is None or len(ia) > len(ib):
                    ia = ia[:-1]
                else:
                    ib = ib[:-1]
            row = [bos] + ia + [eos] + ([eos] + ib + [eos] if ib is not None else [])
            mask_rows.append([1] * len(row) + [0] * (max_length - len(row)))
            ids_rows.append(row + [pad] * (max_length - len(row)))
        return {"input_ids": torch.tensor(ids_rows, dtype=torch.long), "attention_mask": torch.tensor(mask_rows, dtype=torch.long)}


@contextlib.contextmanager
def stubbed(cap):
    import torch
    import transformers
    sys.path.insert(0, REF)
    import mdr.retrieval.utils.utils as ref_utils
    saved_mods = {k: sys.modules.get(k) for k in ("faiss", "apex", "apex.amp", "tqdm")}
    saved = (torch.nn.Module.to, ref_utils.move_to_cuda, transformers.AutoTokenizer, sys.argv)
    apex = types.ModuleType("apex")
    apex.amp = types.ModuleType("apex.amp")
    apex.amp.initialize = lambda model, opt_level="O1": model
    apex.amp.register_half_function = lambda *a, **k: None
    tq = types.ModuleType("tqdm")
    tq.tqdm = lambda it, *a, **k: it
    sys.modules.update({"faiss": faiss_stub(cap), "apex": apex, "apex.amp": apex.amp, "tqdm": tq})
    real_auto = transformers.AutoTokenizer

    class AutoTokenizer211:
        @staticmethod
        def from_pretrained(name, *a, **k):
            return RobertaTokenizer211(real_auto.from_pretrained(name, *a, **k), cap)

    torch.nn.Module.to = lambda self, *a, **k: self
    ref_utils.move_to_cuda = lambda sample: sample
    transformers.AutoTokenizer = AutoTokenizer211
    try:
        yield
    finally:
        torch.nn.Module.to, ref_utils.move_to_cuda, transformers.AutoTokenizer, sys.argv = saved
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.path.remove(REF)


def run_reference(argv):
    """Executes the reference's script as __main__ with `argv`; returns (globals of the run, Capture, its stderr)."""
    import logging
    import torch
    cap = Capture()
    err = io.StringIO()
    root_logger = logging.getLogger()
    keep = (root_logger.level, list(root_logger.handlers))
    with stubbed(cap), contextlib.redirect_stderr(err):  # the script's StreamHandler is created inside: it binds this stderr
        sys.argv = [SCRIPT] + argv
        torch.manual_seed(0)
        g = runpy.run_path(SCRIPT, run_name="__main__")
    root_logger.handlers[:] = keep[1]
    root_logger.setLevel(keep[0])
    return g, cap, err.getvalue()


FEVER_SCRIPT = os.path.join(REF, "scripts", "eval", "eval_mhop_fever.py")
FEVER_CASES = [(2, 5, 6), (3, 2, 4)]  # (--beam-size-1, --beam-size-2, --topk)


def fever_argv(a, b1, b2, topk, save):
    return [a["raw_fever"], a["index"], a["id2doc"]["list"], a["ckpt"], "--batch-size", str(BATCH), "--beam-size-1", str(b1), "--beam-size-2", str(b2),
            "--topk", str(topk), "--model-name", a["model_dir"], "--gpu", "--shared-encoder", "--save-path", save, "--max-q-len", str(MAX_Q_LEN),
            "--max-q-sp-len", str(MAX_Q_SP_LEN)]


def run_reference_fever(argv):
    """The reference's scripts/eval/eval_mhop_fever.py executed as __main__ under the same library stubs. It imports `models.*` / `utils.*` relative to
    mdr/retrieval, and its LAST statement writes to a directory on its author's machine (`/private/home/...`, :172), which fails everywhere else: the script's
    own code object is therefore exec'd in a namespace this function keeps, so that `retrieval_outputs` (:159-169, filled before that write) survives it."""
    import logging
    import torch
    cap = Capture()
    err = io.StringIO()
    root_logger = logging.getLogger()
    keep = (root_logger.level, list(root_logger.handlers))
    pkg = os.path.join(REF, "mdr", "retrieval")
    clash = {k: sys.modules.pop(k) for k in list(sys.modules) if k in ("models", "utils") or k.startswith("models.") or k.startswith("utils.")}
    sys.path.insert(0, pkg)
    ns = {"__name__": "__main__", "__file__": FEVER_SCRIPT}
    try:
        with stubbed(cap), contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
            import utils.utils as fever_utils  # the module object THIS script imports move_to_cuda from
            fever_utils.move_to_cuda = lambda sample: sample
            sys.argv = [FEVER_SCRIPT] + argv
            torch.manual_seed(0)
            try:
                exec(compile(open(FEVER_SCRIPT).read(), FEVER_SCRIPT, "exec"), ns)
                raise AssertionError("the hard-coded output directory exists here?")
            except FileNotFoundError as e:
                assert "/private/home" in str(e), e
    finally:
        sys.path.remove(pkg)
        for k in [k for k in sys.modules if k in ("models", "utils") or k.startswith("models.") or k.startswith("utils.")]:
            del sys.modules[k]
        sys.modules.update(clash)
        root_logger.handlers[:] = keep[1]
        root_logger.setLevel(keep[0])
    return ns, cap, err.getvalue()


ENCODE_SCRIPT = os.path.join(REF, "scripts", "encode_corpus.py")
ENCODE_MAX_C_LEN, ENCODE_BATCH = 30, 50
ENCODE_ROWS = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 49, 50, 51, 77, 99, 100, 130, 150, 199, 200, 201, 255, 256)  # embedding rows kept in the fixture (5, 77, 130, 200: empty text)


ENCODE_MAX_Q_LEN = 14  # --max_q_len of the --is_query_embed run (what its sequences are cut at)
# the branches of EmDataset no fixture executed before round 6: (name, asset key of --predict_file, extra flags)
ENCODE_VARIANTS = [("tsv", "corpus_tsv", []), ("fever", "corpus_fever", []), ("query_embed", "queries_embed", ["--is_query_embed", "--max_q_len", str(ENCODE_MAX_Q_LEN)])]


def encode_argv(a, save, predict_file=None, extra=()):
    return ["--do_predict", "--predict_batch_size", str(ENCODE_BATCH), "--model_name", a["model_dir"], "--predict_file", predict_file or a["corpus_jsonl"],
            "--init_checkpoint", a["ckpt"], "--embed_save_path", save, "--fp16", "--max_c_len", str(ENCODE_MAX_C_LEN), "--num_workers", "0"] + list(extra)


def run_reference_encode_corpus(argv):
    """The reference's scripts/encode_corpus.py executed as __main__ under the same stubs (no GPU here: it picks the CPU itself, :50-52; `--fp16` reaches
    the apex stub). Returns its stdout (EmDataset's prints, `embeds.size()`)."""
    cap = Capture()
    out = io.StringIO()
    with stubbed(cap), contextlib.redirect_stdout(out), contextlib.redirect_stderr(io.StringIO()):
        import torch
        sys.argv = [ENCODE_SCRIPT] + argv
        torch.manual_seed(0)
        runpy.run_path(ENCODE_SCRIPT, run_name="__main__")
    return out.getvalue(), cap


def main():
    tmp = tempfile.mkdtemp(prefix="mdr_cli_golden_")
    a = build_assets(tmp)
    import transformers
    # AutoModel.from_pretrained(--model-name) (mhop_retriever.py:20) wants weights next to the config; load_saved then overwrites every one of them
    transformers.RobertaModel(transformers.AutoConfig.from_pretrained(a["model_dir"])).save_pretrained(a["model_dir"])
    build_base12_assets(a)
    transformers.RobertaModel(transformers.AutoConfig.from_pretrained(a["model_dir_base12"])).save_pretrained(a["model_dir_base12"])
    meta = {"n_docs": N_DOCS, "n_questions": N_Q, "seed": SEED, "batch": BATCH, "max_q_len": MAX_Q_LEN, "max_q_sp_len": MAX_Q_SP_LEN,
            "generator": "oracle/gen_cli_golden.py: /root/reference/scripts/eval/eval_mhop_retrieval.py executed as __main__ under library stubs",
            "cases": []}
    arrays = {}
    for ci, (beam, topk, shape, extra) in enumerate(CASES):
        save = os.path.join(tmp, f"paths_{ci}.jsonl")
        g, cap, err = run_reference(cli_argv(a, beam, topk, shape, extra, save))
        n_batches = -(-len(g["questions"]) // BATCH)
        assert len(cap.searches) == 2 * n_batches and len(cap.tokenizer_calls) == 2 * n_batches
        jsonl = open(save, "rb").read().decode("utf-8")
        log = [ln for ln in err.split("\n") if ln != "" and "Loading weights" not in ln]  # (transformers' own progress bar is not the script's output)
        case = {"beam": beam, "topk": topk, "id2doc_shape": shape, "extra_flags": extra, "n_batches": n_batches, "log": log,
                "metrics": g["metrics"], "questions_encoded": g["questions"], "jsonl_sha256": hashlib.sha256(jsonl.encode()).hexdigest(),
                "jsonl": jsonl if beam <= 5 else None, "hop2_pairs": [cap.tokenizer_calls[2 * b + 1] for b in range(n_batches)] if beam <= 5 else None,
                "hop2_pairs_sha256": hashlib.sha256(json.dumps([cap.tokenizer_calls[2 * b + 1] for b in range(n_batches)]).encode()).hexdigest(),                }
        # per question the (hop-1 title, hop-2 title) pairs of its record: titles are what the reference's metrics work on (:219-242)
        case["chain_titles"] = [[[c[0]["title"], c[1]["title"]] for c in json.loads(ln)["candidate_chains"]] for ln in jsonl.split("\n") if ln]
        n_inf = 0
        for b in range(n_batches):
            h1, h2 = cap.searches[2 * b], cap.searches[2 * b + 1]
            arrays[f"c{ci}.b{b}.D"], arrays[f"c{ci}.b{b}.I"] = h1["D"], h1["I"].astype(np.int32)
            arrays[f"c{ci}.b{b}.D2"], arrays[f"c{ci}.b{b}.I2"] = h2["D"], h2["I"].astype(np.int32)
            if ci == 1:  # the embeddings the script searched with, once (hop 1 is the same in every case): the GPU test feeds them to the HIP index
                arrays[f"c{ci}.b{b}.q"], arrays[f"c{ci}.b{b}.q2"] = h1["x"], h2["x"]
            n_inf += sum(int(i) in EMPTY_DOCS for i in h1["I"].ravel())
        case["empty_passages_in_hop1_beams"] = n_inf
        meta["cases"].append(case)
        print(f"case {ci}: beam {beam} topk {topk} {shape} {extra}: {len(g['metrics'])} metrics, {len(jsonl)} JSONL bytes, "
              f"{n_inf} empty passages in hop-1 beams, log tail: {log[-1]!r}")
    assert any(c["empty_passages_in_hop1_beams"] for c in meta["cases"]), "no case exercised the empty-passage rule: change the seed"
    # ---- the FEVER variant (scripts/eval/eval_mhop_fever.py: separate hop widths, list-valued corpus dict, (title, text) chains looked up BY TITLE) ----
    meta["fever_cases"] = []
    for fi, (b1, b2, topk) in enumerate(FEVER_CASES):
        ns, cap, err = run_reference_fever(fever_argv(a, b1, b2, topk, "fever_out.jsonl"))
        n_batches = -(-len(ns["questions"]) // BATCH)
        assert len(cap.searches) == 2 * n_batches and len(ns["retrieval_outputs"]) == len(a["claims"])
        lines = [json.dumps(r) for r in ns["retrieval_outputs"]]
        meta["fever_cases"].append({"beam1": b1, "beam2": b2, "topk": topk, "n_batches": n_batches, "jsonl": "".join(ln + "\n" for ln in lines),
                                    "hop2_pairs": [cap.tokenizer_calls[2 * b + 1] for b in range(n_batches)],
                                    "log": [ln for ln in err.split("\n") if ln != "" and "Loading weights" not in ln]})
        n_inf = 0
        for b in range(n_batches):
            h1, h2 = cap.searches[2 * b], cap.searches[2 * b + 1]
            arrays[f"f{fi}.b{b}.D"], arrays[f"f{fi}.b{b}.I"] = h1["D"], h1["I"].astype(np.int32)
            arrays[f"f{fi}.b{b}.D2"], arrays[f"f{fi}.b{b}.I2"] = h2["D"], h2["I"].astype(np.int32)
            n_inf += sum(int(i) in EMPTY_DOCS for i in h1["I"].ravel())
        print(f"fever case {fi}: beam {b1} x {b2} topk {topk}: {len(lines)} records, {sum(map(len, lines))} JSONL bytes, {n_inf} empty passages in hop-1 beams")
    # ---- the corpus encoder (scripts/encode_corpus.py: EmDataset -> DataLoader(em_collate) -> RobertaCtxEncoder -> np.save + id2doc.json) ----
    save = os.path.join(tmp, "emb")
    stdout, ecap = run_reference_encode_corpus(encode_argv(a, save))
    emb = np.load(save + ".npy")
    assert emb.shape == (N_DOCS, 768) and emb.dtype == np.float32
    meta["encode_corpus"] = {"max_c_len": ENCODE_MAX_C_LEN, "predict_batch_size": ENCODE_BATCH, "rows": list(ENCODE_ROWS), "shape": list(emb.shape),
                             "id2doc_json": open(os.path.join(save, "id2doc.json")).read(), "embeddings_sha256": hashlib.sha256(emb.tobytes()).hexdigest(),
                             "stdout": [ln.replace(tmp, "<assets>") for ln in stdout.split("\n") if ln and "it/s]" not in ln and not ln.startswith("\r")]}
    arrays["encode.rows"] = emb[list(ENCODE_ROWS)]
    arrays["encode.norms"] = np.linalg.norm(emb, axis=1).astype(np.float32)
    # what the reference's EmDataset handed to its tokenizer for the passages with special titles / texts: [title argument, text_pair argument, max_length]
    meta["encode_corpus"]["encode_plus_args"] = {str(i): ecap.encode_plus_calls[i] for i in sorted(set(TITLE_OVERRIDES) | set(TEXT_SUFFIXES) | set(EMPTY_DOCS))}
    assert len(ecap.encode_plus_calls) == N_DOCS
    print(f"encode_corpus: {emb.shape}, stdout {meta['encode_corpus']['stdout']}")
    # ---- EmDataset's other branches (encode_datasets.py:52-72): TSV reader, "fever" in the path, --is_query_embed ----
    meta["encode_variants"] = {}
    for name, key, extra in ENCODE_VARIANTS:
        vsave = os.path.join(tmp, "emb_" + name)
        vout, vcap = run_reference_encode_corpus(encode_argv(a, vsave, a[key], extra))
        vemb = np.load(vsave + ".npy")
        idp = os.path.join(vsave, "id2doc.json")
        meta["encode_variants"][name] = {
            "predict_file": os.path.basename(a[key]), "extra_flags": extra, "shape": list(vemb.shape),
            "id2doc_json_sha256": hashlib.sha256(open(idp, "rb").read()).hexdigest() if os.path.exists(idp) else None,
            "id2doc_written": os.path.exists(idp), "save_dir_created": os.path.isdir(vsave),
            "embeddings_equal_jsonl_run": bool(vemb.shape == emb.shape and np.array_equal(vemb, emb)),
            "encode_plus_args_sha256": hashlib.sha256(json.dumps(vcap.encode_plus_calls).encode()).hexdigest(),
            "encode_plus_args_head": vcap.encode_plus_calls[:3], "n_items": len(vcap.encode_plus_calls),
            "stdout": [ln.replace(tmp, "<assets>") for ln in vout.split("\n") if ln and "it/s]" not in ln and not ln.startswith("\r")]}
        arrays[f"encode.{name}.rows"] = vemb[:8]
        print(f"encode_corpus [{name}]: {vemb.shape}, id2doc written {os.path.exists(idp)}, equal to the JSONL run {meta['encode_variants'][name]['embeddings_equal_jsonl_run']}")
    # ---- --topk > beam^2 (eval_mhop_retrieval.py:197-198: ranked_pairs has beam^2 rows) ----
    try:
        run_reference(cli_argv(a, 2, 5, "list", [], os.path.join(tmp, "paths_topk_too_large.jsonl")))
        meta["topk_exceeds_beam_squared"] = {"beam": 2, "topk": 5, "raised": None}
    except Exception as e:  # noqa: BLE001 -- the TYPE is the captured fact
        meta["topk_exceeds_beam_squared"] = {"beam": 2, "topk": 5, "raised": type(e).__name__, "message": str(e)}
    print("topk > beam^2:", meta["topk_exceeds_beam_squared"])
    with open(os.path.join(GOLD, "cli_ref.json"), "w") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)
    np.savez_compressed(os.path.join(GOLD, "cli_ref.npz"), **arrays)
    print("wrote", os.path.join(GOLD, "cli_ref.json"), os.path.getsize(os.path.join(GOLD, "cli_ref.json")), "bytes;",
          os.path.join(GOLD, "cli_ref.npz"), os.path.getsize(os.path.join(GOLD, "cli_ref.npz")), "bytes")


if __name__ == "__main__":
    main()

Tokenizer211 = RobertaTokenizer211  # (name used by the tests)

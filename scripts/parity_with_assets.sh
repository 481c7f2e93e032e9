#!/bin/bash
# One command that makes the FIRST run with the real assets decisive. It cannot run in the build container (no
# network, no HotpotQA files, no transformers==2.11.0); on a machine that has them:
#
#   scripts/parity_with_assets.sh DATA_DIR MODELS_DIR ROBERTA_DIR [PY211]
#
#   DATA_DIR     data/ as /root/reference/scripts/download_hotpot.sh leaves it (hotpot/hotpot_qas_val.json, hotpot_index/wiki_index.npy,
#                hotpot_index/wiki_id2doc.json)
#   MODELS_DIR   models/ of the same script (q_encoder.pt)
#   ROBERTA_DIR  a local roberta-base directory (config.json, vocab.json, merges.txt)
#   PY211        optional: a python interpreter whose environment has transformers==2.11.0 (the reference's pin, requirements.txt:1);
#                without it step 1 is skipped and says so
#
# Step 1  tokenisation: PY211 dumps the token ids transformers 2.11 produces for a fixed set of questions, (question, passage) pairs and
#         (title, text) passages -- the three call sites of the reference (eval_mhop_retrieval.py:148,168, encode_datasets.py:95) -- into
#         tests/golden/tokenizer_2_11.json; this build's restatement (data.tokenize_2_11 / encode_pairs_2_11 / prefix_space_2_11, written
#         from memory of 2.11 and UNPINNED until this runs) is then checked against it token for token.
# Step 2  retrieval quality: the drop-in CLI on HotpotQA dev, beam 1 / topk 1 (the README command), and its log lines are compared with
#         /root/reference/README.md:77-91 within +-0.002.
# Exit code 0 = both checks passed (or step 1 skipped and step 2 passed); 1 = a mismatch, printed.
set -eu
DATA=${1:?data dir}; MODELS=${2:?models dir}; ROBERTA=${3:?roberta-base dir}; PY211=${4:-}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
OUT=${OUT:-/tmp/mdr_parity}; mkdir -p "$OUT"
GOLD=tests/golden/tokenizer_2_11.json

if [ -n "$PY211" ]; then
  echo "== step 1: token ids of transformers 2.11 -> $GOLD"
  "$PY211" - "$DATA" "$ROBERTA" "$GOLD" <<'PY'
import json, sys
import transformers
assert transformers.__version__.startswith("2.11"), transformers.__version__
from transformers import AutoTokenizer
data, roberta, gold = sys.argv[1:4]
tok = AutoTokenizer.from_pretrained(roberta)
qs = [json.loads(l) for l in open(f"{data}/hotpot/hotpot_qas_val.json")][:200]
questions = [q["question"][:-1] if q["question"].endswith("?") else q["question"] for q in qs]
id2doc = json.load(open(f"{data}/hotpot_index/wiki_id2doc.json"))
docs = [id2doc[str(i)] for i in list(range(100)) + list(range(100000, 100100))]
docs = [{"title": d[0], "text": d[1]} if isinstance(d, list) else d for d in docs]
out = {"version": transformers.__version__, "questions": questions, "docs": docs}
e = tok.batch_encode_plus(questions, max_length=70, pad_to_max_length=True, return_tensors="pt")                       # eval_mhop_retrieval.py:148
out["hop1"] = {"input_ids": e["input_ids"].tolist(), "attention_mask": e["attention_mask"].tolist()}
pairs = [(questions[i], docs[i]["text"] if docs[i]["text"].strip() else docs[i]["title"]) for i in range(200)]
e = tok.batch_encode_plus(pairs, max_length=350, pad_to_max_length=True, return_tensors="pt")                          # :168
out["hop2"] = {"input_ids": e["input_ids"].tolist(), "attention_mask": e["attention_mask"].tolist()}
e = tok.batch_encode_plus(pairs, max_length=351, pad_to_max_length=True, return_tensors="pt")                          # odd budget: the slow truncation rule
out["hop2_odd"] = {"input_ids": e["input_ids"].tolist()}
out["ctx"] = [tok.encode_plus(d["title"].strip(), text_pair=(d["text"].strip() or d["title"]), max_length=300)["input_ids"] for d in docs]  # encode_datasets.py:95
# round 6 (VERDICT r5 What's weak 1): the strings whose 2.11 treatment the build could not confirm from memory -- a trailing blank left by the "?" strip (does
# tokenize()'s split_on_tokens strip it?), leading blanks, the tokenizer's own special-token strings inside a text, precomposed / decomposed titles
import unicodedata
probe_q = ["When was the stadium born ", "  where is the river bank", "what is <mask> in <s> text </s> here", "plain question", "tab\tinside and trailing tab\t",
           "Who founded Zu\u0308rich", "Who founded Z\u00fcrich"]
probe_p = [("Z\u00fcrich", "Z\u00fcrich is a city <s> inner </s> <mask> tail"), (unicodedata.normalize("NFD", "Krak\u00f3w"), "so <mask> it </s>"),
           ("  Krak\u00f3w  ", " leading blank and trailing blank "), ("\u00c5ngstr\u00f6m (unit)", "<s>"), ("T", "")]
out["probe_questions"], out["probe_passages"] = probe_q, [list(x) for x in probe_p]
e = tok.batch_encode_plus(probe_q, max_length=70, pad_to_max_length=True, return_tensors="pt")
out["probe_hop1"] = e["input_ids"].tolist()
pp = [(probe_q[i % len(probe_q)], t if t.strip() else ti) for i, (ti, t) in enumerate(probe_p)]
e = tok.batch_encode_plus(pp, max_length=40, pad_to_max_length=True, return_tensors="pt")
out["probe_hop2"] = e["input_ids"].tolist()
out["probe_ctx"] = [tok.encode_plus(unicodedata.normalize("NFD", ti.strip()), text_pair=(t.strip() or ti), max_length=30)["input_ids"] for ti, t in probe_p]  # encode_datasets.py:95
json.dump(out, open(gold, "w"))
print("wrote", gold)
PY
  python - "$ROBERTA" "$GOLD" <<'PY'
import json, sys
import numpy as np
from transformers import AutoTokenizer
from multihop_dense_retrieval_amd.data import encode_pairs_2_11, tokenize_2_11
roberta, gold = sys.argv[1:3]
g = json.load(open(gold))
tok = AutoTokenizer.from_pretrained(roberta)
bad = 0
e = tokenize_2_11(tok, g["questions"], None, 70)
bad += int((np.asarray(e["input_ids"]) != np.asarray(g["hop1"]["input_ids"])).any(1).sum())
pairs = [(g["questions"][i], g["docs"][i]["text"] if g["docs"][i]["text"].strip() else g["docs"][i]["title"]) for i in range(200)]
for L, key in ((350, "hop2"), (351, "hop2_odd")):
    e = tokenize_2_11(tok, None, pairs, L)
    n = int((np.asarray(e["input_ids"]) != np.asarray(g[key]["input_ids"])).any(1).sum())
    print(f"{key}: {n} of 200 rows differ")
    bad += n
ids, _ = encode_pairs_2_11(tok, [d["title"].strip() for d in g["docs"]], [(d["text"].strip() or d["title"]) for d in g["docs"]], 300, False)
n = sum(a != b for a, b in zip(ids, g["ctx"]))
print(f"ctx: {n} of {len(ids)} passages differ")
bad += n
if "probe_questions" in g:  # the strings of round 6 (trailing / leading blanks, special-token strings, NFD titles): reported one by one
    import unicodedata
    pq, pp_ = g["probe_questions"], g["probe_passages"]
    e = tokenize_2_11(tok, pq, None, 70)
    for i, (a, b) in enumerate(zip(np.asarray(e["input_ids"]).tolist(), g["probe_hop1"])):
        if a != b:
            bad += 1
            print(f"probe question {pq[i]!r}: ours {[t for t in a if t != 1]} vs 2.11 {[t for t in b if t != 1]}")
    pairs = [(pq[i % len(pq)], t if t.strip() else ti) for i, (ti, t) in enumerate(pp_)]
    e = tokenize_2_11(tok, None, pairs, 40)
    for i, (a, b) in enumerate(zip(np.asarray(e["input_ids"]).tolist(), g["probe_hop2"])):
        if a != b:
            bad += 1
            print(f"probe pair {pairs[i]!r}: ours {[t for t in a if t != 1]} vs 2.11 {[t for t in b if t != 1]}")
    ids, _ = encode_pairs_2_11(tok, [unicodedata.normalize("NFD", ti.strip()) for ti, t in pp_], [(t.strip() or ti) for ti, t in pp_], 30, False)
    for i, (a, b) in enumerate(zip(ids, g["probe_ctx"])):
        if list(a) != list(b):
            bad += 1
            print(f"probe passage {pp_[i]!r}: ours {list(a)} vs 2.11 {list(b)}")
print("tokenisation parity with transformers", g["version"], "OK" if bad == 0 else f"FAILED ({bad} rows)")
if bad:  # the two open rules are one-line switches in data.py: say which combination reproduces 2.11's ids
    from multihop_dense_retrieval_amd import data as _d

    def mismatches():
        n = int((np.asarray(tokenize_2_11(tok, g["questions"], None, 70)["input_ids"]) != np.asarray(g["hop1"]["input_ids"])).any(1).sum())
        pr = [(g["questions"][i], g["docs"][i]["text"] if g["docs"][i]["text"].strip() else g["docs"][i]["title"]) for i in range(200)]
        n += int((np.asarray(tokenize_2_11(tok, None, pr, 350)["input_ids"]) != np.asarray(g["hop2"]["input_ids"])).any(1).sum())
        if "probe_questions" in g:
            n += sum(a != b for a, b in zip(np.asarray(tokenize_2_11(tok, g["probe_questions"], None, 70)["input_ids"]).tolist(), g["probe_hop1"]))
        return n
    keep = (_d.PREFIX_SPACE_2_11, _d.RSTRIP_SEGMENTS_2_11)
    for ps in (True, False):
        for rs in (False, True):
            _d.PREFIX_SPACE_2_11, _d.RSTRIP_SEGMENTS_2_11 = ps, rs
            print(f"  data.PREFIX_SPACE_2_11 = {ps}, data.RSTRIP_SEGMENTS_2_11 = {rs}: {mismatches()} rows differ from transformers {g['version']}")
    _d.PREFIX_SPACE_2_11, _d.RSTRIP_SEGMENTS_2_11 = keep
sys.exit(0 if bad == 0 else 1)
PY
else
  echo "== step 1 SKIPPED: no transformers==2.11.0 interpreter given (tokenisation fidelity stays unpinned)"
fi

echo "== step 2: the README command on HotpotQA dev"
python scripts/eval/eval_mhop_retrieval.py "$DATA/hotpot/hotpot_qas_val.json" "$DATA/hotpot_index/wiki_index.npy" "$DATA/hotpot_index/wiki_id2doc.json" \
  "$MODELS/q_encoder.pt" --batch-size 100 --beam-size 1 --topk 1 --shared-encoder --model-name "$ROBERTA" --gpu --save-path "$OUT/paths_top1.jsonl" \
  2> "$OUT/eval.log" || { tail -20 "$OUT/eval.log"; exit 1; }
python - "$OUT/eval.log" <<'PY'
import re, sys
# /root/reference/README.md:77-91
want = {"all": (7405, 0.8428089128966915, 0.6592842673869007, 0.7906819716407832, 0.6592842673869007),
        "comparison": (1487, 0.9932750504371217, 0.9482178883658372, 0.9643577673167452, 0.9482178883658372),
        "bridge": (5918, 0.805001689760054, 0.5866846907739101, 0.7470429199053734, 0.5866846907739101)}
log = open(sys.argv[1]).read()
blocks, cur = {}, None
for line in log.splitlines():
    m = re.match(r"Evaluating (\d+) samples", line)
    if m:
        cur = "all"; blocks[cur] = [int(m.group(1))]; continue
    m = re.match(r"(\w+) Questions num: (\d+)", line)
    if m:
        cur = m.group(1); blocks[cur] = [int(m.group(2))]; continue
    m = re.match(r"\t(Avg PR|Avg P-EM|Avg 1-Recall|Path Recall): ([0-9.eE+-]+)", line)
    if m and cur:
        blocks[cur].append(float(m.group(2)))
ok = True
for k, w in want.items():
    got = blocks.get(k)
    if not got or len(got) != 5:
        print(k, "MISSING in the log"); ok = False; continue
    good = got[0] == w[0] and all(abs(a - b) <= 0.002 for a, b in zip(got[1:], w[1:]))
    print(f"{k:10s} n {got[0]} (want {w[0]})  PR {got[1]:.4f}/{w[1]:.4f}  P-EM {got[2]:.4f}/{w[2]:.4f}  1-Recall {got[3]:.4f}/{w[3]:.4f}  Path {got[4]:.4f}/{w[4]:.4f}  {'ok' if good else 'MISMATCH'}")
    ok &= good
print("README parity (+-0.002):", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
PY

"""Answer-string recall of the `--only-eval-ans` mode of scripts/eval/eval_mhop_retrieval.py: CPU string matching on the
retrieved passages, no kernel involved.

  para_has_answer      /root/reference/mdr/retrieval/utils/utils.py:126-139
  simple_words         the `words(uncased=True)` view of SimpleTokenizer.tokenize,
                       /root/reference/mdr/retrieval/utils/basic_tokenizer.py:238-277 (+ :46-55)
  chain_text           the passage concatenation at /root/reference/scripts/eval/eval_mhop_retrieval.py:210-212
  answer_summary_lines the log lines at :269-273

Pinned by tests/golden/answer_recall.json, produced by the reference functions themselves (oracle/gen_golden.py).
"""
import unicodedata

import numpy as np
import regex

# alphanumeric runs (letters, numbers, combining marks) or any single character that is neither a separator nor a control
_TOKEN = regex.compile(r"([\p{L}\p{N}\p{M}]+)|([^\p{Z}\p{C}])", flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)


def simple_words(text):
    """Lower-cased tokens of `text` (no normalisation here)."""
    return [m.group().lower() for m in _TOKEN.finditer(text)]


def para_has_answer(answers, para):
    """True when the token sequence of any answer occurs contiguously in the paragraph's (NFD-normalised, lower-cased)."""
    if not isinstance(answers, list):
        raise AssertionError("answer must be a list of strings")
    text = simple_words(unicodedata.normalize("NFD", para))
    for ans in answers:
        a = simple_words(unicodedata.normalize("NFD", ans))
        for i in range(0, len(text) - len(a) + 1):  # an answer with no tokens matches at once, like the reference
            if a == text[i:i + len(a)]:
                return True
    return False


def chain_text(chains, id2doc):
    """"yes no " followed by `title text` of both passages of every chain; chains follow each other WITHOUT a separator."""
    out = "yes no "
    for h1, h2, _ in chains:
        out += " ".join(id2doc[str(d)]["title"] + " " + id2doc[str(d)]["text"] for d in (h1, h2))
    return out


def answer_metrics(item, chains, id2doc):
    return {"question": item["question"], "ans_recall": int(para_has_answer(item["answer"], chain_text(chains, id2doc))),
            "type": item.get("type", "single")}


def answer_summary_lines(metrics):
    groups = {}
    for m in metrics:
        groups.setdefault(m["type"], []).append(m)
    lines = [f"Evaluating {len(metrics)} samples...", f'Ans Recall: {np.mean([m["ans_recall"] for m in metrics])}']
    for t, ms in groups.items():
        lines.append(f"{t} Questions num: {len(ms)}")
        lines.append(f'Ans Recall: {np.mean([m["ans_recall"] for m in ms])}')
    return lines

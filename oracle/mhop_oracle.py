"""oracle/mhop_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-Python/numpy restatement of the host-side two-hop logic that sits inline under `__main__` in
/root/reference/scripts/eval/eval_mhop_retrieval.py (it cannot be imported: top-level `import faiss`,
`from apex import amp`, no functions). Each function names the lines it follows.

Only tests/ may import this module.
"""
import collections

import numpy as np


def strip_question(q):
    """eval_mhop_retrieval.py:139 -- exactly one trailing '?' removed for encoding."""
    return q[:-1] if q.endswith("?") else q


def normalise_id2doc(id2doc):
    """eval_mhop_retrieval.py:131-133 -- list-valued corpus dict -> {"title","text"} dicts."""
    if isinstance(id2doc["0"], list):
        return {k: {"title": v[0], "text": v[1]} for k, v in id2doc.items()}
    return id2doc


def build_hop2_pairs(batch_q, D, I, id2doc, roberta=True):
    """eval_mhop_retrieval.py:158-166 -- (question, passage text) pairs for hop 2, row-major over
    (question, beam slot); an empty passage falls back to its title and its hop-1 score becomes -inf
    (D is modified in place, as the reference does)."""
    pairs = []
    for b in range(len(batch_q)):
        for j, doc_id in enumerate(I[b]):
            doc = id2doc[str(doc_id)]["text"]
            if roberta and doc.strip() == "":
                doc = id2doc[str(doc_id)]["title"]
                D[b][j] = float("-inf")
            pairs.append((batch_q[b], doc))
    return pairs


def rank_paths(D, I, D2, I2, beam, topk):
    """eval_mhop_retrieval.py:181-206 -- path score = hop-1 score + hop-2 score over the beam x beam
    grid; per question the `topk` best (hop1 id, hop2 id) pairs, best first.

    The reference sorts with np.argsort(...)[::-1] (unstable introsort, reversed), so the order among
    EQUAL path scores is unspecified there; this restatement returns that exact numpy expression's
    order, and tests only compare tie-free cases plus set-equality on tied ones."""
    bsize = D.shape[0]
    D2 = D2.reshape(bsize, beam, beam)
    I2 = I2.reshape(bsize, beam, beam)
    path_scores = np.expand_dims(D, axis=2) + D2
    out = []
    for b in range(bsize):
        flat = path_scores[b].ravel()
        order = np.argsort(flat)[::-1]
        ij = np.vstack(np.unravel_index(order, (beam, beam))).transpose()
        chains = []
        for r in range(topk):  # IndexError when topk > beam*beam, as in the reference (:197-198)
            i, j = ij[r]
            chains.append((int(I[b, i]), int(I2[b, i, j]), float(flat[order[r]])))
        out.append(chains)
    return out


def question_metrics(chains, sp, id2doc):
    """eval_mhop_retrieval.py:219-242 -- title-based p_recall / p_em / recall_1 / path_covered."""
    assert len(set(sp)) == 2
    retrieved, hop1, path_titles = [], [], []
    for h1, h2, _ in chains:
        t1, t2 = id2doc[str(h1)]["title"], id2doc[str(h2)]["title"]
        retrieved += [t1, t2]
        hop1.append(t1)
        path_titles.append([t1, t2])
    covered = [t in retrieved for t in sp]
    return {
        "p_recall": int(np.sum(covered) > 0),
        "p_em": int(np.sum(covered) == len(covered)),
        "recall_1": int(np.sum([t in hop1 for t in sp]) > 0),
        "path_covered": int(np.sum([int(set(p) == set(sp)) for p in path_titles]) > 0),
    }


def output_record(item, chains, id2doc):
    """eval_mhop_retrieval.py:246-258 -- JSONL record; key order _id, question, candidate_chains;
    the question keeps its original '?'; chain elements are the id2doc values verbatim."""
    return {
        "_id": item["_id"],
        "question": item["question"],
        "candidate_chains": [[id2doc[str(h1)], id2doc[str(h2)]] for h1, h2, _ in chains],
    }


def summary_lines(metrics):
    """eval_mhop_retrieval.py:265-284 -- the exact log lines (non --only-eval-ans branch)."""
    lines = [f"Evaluating {len(metrics)} samples..."]
    by_type = collections.defaultdict(list)
    for m in metrics:
        by_type[m["type"]].append(m)

    def block(ms):
        return [f'\tAvg PR: {np.mean([m["p_recall"] for m in ms])}',
                f'\tAvg P-EM: {np.mean([m["p_em"] for m in ms])}',
                f'\tAvg 1-Recall: {np.mean([m["recall_1"] for m in ms])}',
                f'\tPath Recall: {np.mean([m["path_covered"] for m in ms])}']

    lines += block(metrics)
    for t in by_type.keys():
        lines.append(f"{t} Questions num: {len(by_type[t])}")
        lines += block(by_type[t])
    return lines

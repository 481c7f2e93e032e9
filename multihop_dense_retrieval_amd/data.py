"""Corpus-side data plumbing of scripts/encode_corpus.py:
    collate_tokens   /root/reference/mdr/retrieval/data/data_utils.py:11-29
    EmDataset        /root/reference/mdr/retrieval/data/encode_datasets.py:32-100
    em_collate       /root/reference/mdr/retrieval/data/encode_datasets.py:102-114
Same names and behaviour. The reference was written against transformers 2.11 (`encode_plus(a, text_pair=b,
max_length=n)`: special tokens added, longest-first truncation by the slow tokenizer's pop loop, no padding); the pair
template and that truncation rule are restated here (encode_pairs_2_11) so the result does not depend on the installed
HF version, whose fast tokenizers split an odd token budget differently."""
import csv
import json
import os
import unicodedata

import torch


def truncate_longest_first_2_11(la, lb, budget):
    """Lengths left by transformers 2.11 `truncate_sequences(..., 'longest_first')` (the slow-tokenizer loop the reference
    runs, requirements.txt:1): tokens are popped one at a time from the LONGER sequence, from the SECOND on a tie, until the
    pair fits `budget` = max_length - 4 special tokens. Closed form of that loop. (The Rust fast tokenizers split an odd
    budget the other way round -- floor to the shorter/first, ceil to the longer/second -- so for odd budgets their pair
    encoding differs from the reference's by the boundary token; tests/test_tokenizer_fidelity.py pins both facts.)"""
    over = la + lb - budget
    if over <= 0:
        return la, lb
    d = abs(la - lb)
    if over <= d:  # only the longer sequence is cut
        return (la - over, lb) if la > lb else (la, lb - over)
    s, r = min(la, lb), over - d  # both at length s; the remaining r pops alternate, second sequence first
    return s - r // 2, s - (r + 1) // 2


# transformers 2.11 `RobertaTokenizer.prepare_for_tokenization(text, add_special_tokens=False, **kw)` prepends ONE space to a text that
# does not start with whitespace whenever `add_prefix_space` (default: = add_special_tokens) is on, and `encode_plus` /
# `batch_encode_plus` call `tokenize(text, add_special_tokens=True)` for EVERY segment -- so under the reference's pin
# (requirements.txt:1) the question, the title and the passage text each start with a "G-dot" (space-prefixed) BPE token. From 3.0 on
# add_prefix_space is a constructor argument that defaults to False, which is what the installed tokenizer does. Restated from the
# 2.11 source as remembered (the package is not installable offline: UNPINNED); one switch for every call site.
PREFIX_SPACE_2_11 = True


def is_roberta_family(tokenizer):
    return "Roberta" in tokenizer.__class__.__name__


def prefix_space_2_11(text):
    """The text as transformers 2.11's RoBERTa tokenizer sees it inside encode_plus (see PREFIX_SPACE_2_11)."""
    if PREFIX_SPACE_2_11 and text and not text[0].isspace():
        return " " + text
    return text


def encode_pairs_2_11(tokenizer, firsts, seconds, max_length, pad_to_max_length):
    """`encode_plus(a, text_pair=b, max_length=n[, pad_to_max_length=True])` of transformers 2.11 for RoBERTa-family
    tokenizers on top of ANY HF tokenizer version: the tokenizer only supplies the BPE (each text tokenised on its own with
    2.11's prefix space, no special tokens, no truncation); the pair template `<s> A </s></s> B </s>`, the reference's
    truncation rule and the right-padding are applied here. Returns (list of id lists, list of mask lists)."""
    if not is_roberta_family(tokenizer):
        raise TypeError("encode_pairs_2_11 restates the RoBERTa pair template; other tokenizer families use their own call")
    ta = tokenizer([prefix_space_2_11(t) for t in firsts], add_special_tokens=False, truncation=False)["input_ids"]
    tb = tokenizer([prefix_space_2_11(t) for t in seconds], add_special_tokens=False, truncation=False)["input_ids"]
    bos, eos, pad = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
    ids_out, mask_out = [], []
    for a, b in zip(ta, tb):
        na, nb = truncate_longest_first_2_11(len(a), len(b), max_length - 4)
        ids = [bos] + list(a[:max(na, 0)]) + [eos, eos] + list(b[:max(nb, 0)]) + [eos]
        mask = [1] * len(ids)
        if pad_to_max_length and len(ids) < max_length:
            mask += [0] * (max_length - len(ids))
            ids += [pad] * (max_length - len(ids))
        ids_out.append(ids)
        mask_out.append(mask)
    return ids_out, mask_out


def tokenize_2_11(tokenizer, texts, pairs, max_length):
    """`batch_encode_plus(x, max_length=n, pad_to_max_length=True, return_tensors="pt")` of transformers 2.11
    (eval_mhop_retrieval.py:148,168): `<s> q </s>` / `<s> q </s></s> d </s>`, longest-first truncation, right-pad to
    max_length. RoBERTa-family tokenizers (the reference's path): single texts go through the installed tokenizer's own call
    with 2.11's prefix space in front (prefix_space_2_11), pairs through encode_pairs_2_11, which also keeps the
    reference's (slow-tokenizer) truncation rule for odd token budgets. Any other family (the reference's `else` branches for
    BERT-style models): the installed tokenizer's own pair call, `[CLS] a [SEP] b [SEP]` with token_type_ids."""
    if not is_roberta_family(tokenizer):
        if pairs is None:
            return tokenizer(list(texts), max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")
        return tokenizer([p[0] for p in pairs], [p[1] for p in pairs], max_length=max_length, padding="max_length",
                         truncation="longest_first", return_tensors="pt")
    if pairs is None:
        return tokenizer([prefix_space_2_11(t) for t in texts], max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")
    ids, mask = encode_pairs_2_11(tokenizer, [p[0] for p in pairs], [p[1] for p in pairs], max_length, True)
    return {"input_ids": torch.tensor(ids, dtype=torch.int64), "attention_mask": torch.tensor(mask, dtype=torch.int64)}


def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False):
    """List of 1-D tensors -> one right- (or left-) padded 2-D tensor."""
    values = [v.reshape(-1) for v in values]
    width = max(v.numel() for v in values)
    out = values[0].new_full((len(values), width), pad_idx)
    for row, v in zip(out, values):
        n = v.numel()
        dst = row[width - n:] if left_pad else row[:n]
        if move_eos_to_beginning:
            assert v[-1] == eos_idx
            dst[0] = eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return out


def normalize(text):
    return unicodedata.normalize("NFD", text)


class EmDataset(torch.utils.data.Dataset):
    """Reads a JSONL ({"title","text"[,"intro"]}) or TSV (id, text, title) corpus, writes
    <save_path>/id2doc.json = {idx: [title, text, intro]} and yields one tokenised passage per item."""

    def __init__(self, tokenizer, data_path, max_q_len, max_c_len, is_query_embed, save_path, write_id2doc=True):
        super().__init__()
        self.is_query_embed = is_query_embed
        self.tokenizer = tokenizer
        self.max_c_len = max_c_len
        if not os.path.exists(save_path):
            os.makedirs(save_path, exist_ok=True)
        id2doc_path = os.path.join(save_path, "id2doc.json")
        print(f"Loading data from {data_path}")
        if self.is_query_embed:
            with open(data_path) as f:
                self.data = [json.loads(line.strip()) for line in f.readlines()]
        else:
            if data_path.endswith("tsv"):
                self.data = []
                with open(data_path) as tsv:
                    for row in csv.reader(tsv, delimiter="\t"):
                        if row[0] != "id":
                            self.data.append({"id": row[0], "text": row[1], "title": row[2]})
            else:  # JSONL (the reference's "fever" branch reads the same format)
                with open(data_path) as f:
                    self.data = [json.loads(line) for line in f.readlines()]
            print(f"load {len(self.data)} documents...")
            if write_id2doc:  # under torch.distributed.run only rank 0 writes the mapping
                id2doc = {idx: (doc["title"], doc["text"], doc.get("intro", False)) for idx, doc in enumerate(self.data)}
                with open(id2doc_path, "w") as g:
                    json.dump(id2doc, g)
        self.max_len = max_q_len if is_query_embed else max_c_len
        print(f"Max sequence length: {self.max_len}")

    def __getitem__(self, index):
        sample = self.data[index]
        if is_roberta_family(self.tokenizer) and sample["text"].strip() == "":
            print(f"empty doc title: {sample['title']}")
            sample["text"] = sample["title"]
        if is_roberta_family(self.tokenizer):  # the reference's prefix-space and truncation rules, whatever the HF version's are
            ids, mask = encode_pairs_2_11(self.tokenizer, [normalize(sample["title"].strip())], [sample["text"].strip()], self.max_len, False)
            return {"input_ids": torch.tensor(ids, dtype=torch.int64), "attention_mask": torch.tensor(mask, dtype=torch.int64)}
        return self.tokenizer(normalize(sample["title"].strip()), text_pair=sample["text"].strip(), max_length=self.max_len,
                              truncation=True, return_tensors="pt")

    def __len__(self):
        return len(self.data)


def em_collate(samples):
    if len(samples) == 0:
        return {}
    batch = {
        "input_ids": collate_tokens([s["input_ids"].view(-1) for s in samples], 0),
        "input_mask": collate_tokens([s["attention_mask"].view(-1) for s in samples], 0),
    }
    if "token_type_ids" in samples[0]:
        batch["input_type_ids"] = collate_tokens([s["token_type_ids"].view(-1) for s in samples], 0)
    return batch

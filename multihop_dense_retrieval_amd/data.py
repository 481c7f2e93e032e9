"""Corpus-side data plumbing of scripts/encode_corpus.py:
    collate_tokens   /root/reference/mdr/retrieval/data/data_utils.py:11-29
    EmDataset        /root/reference/mdr/retrieval/data/encode_datasets.py:32-100
    em_collate       /root/reference/mdr/retrieval/data/encode_datasets.py:102-114
Same names and behaviour; tokenizer calls use the transformers>=4 spelling of the 2.11 API the
reference was written against (`encode_plus(a, text_pair=b, max_length=n)` == `tok(a, b,
truncation=True, max_length=n)`: longest-first truncation, special tokens added, no padding)."""
import csv
import json
import os
import unicodedata

import torch


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
        if "Roberta" in self.tokenizer.__class__.__name__ and sample["text"].strip() == "":
            print(f"empty doc title: {sample['title']}")
            sample["text"] = sample["title"]
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

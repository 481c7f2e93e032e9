"""Fast on-disk corpus store: the reference parses a multi-GB `id2doc.json` into a 5 M-entry
Python dict at every start (/root/reference/scripts/eval/eval_mhop_retrieval.py:131-135, minutes and > 10 GB of host RAM
on the Wikipedia corpus). The store holds the same data as one memory-mapped file with offset tables; opening it takes
milliseconds and reads only the passages a run touches.

    build_store(id2doc_json_path_or_dict, store_path)       once (the CLIs do it on first use with --corpus-store)
    docs = CorpusStore(store_path); docs["42"]["title"]      read-only Mapping: keys "0" .. str(n-1), values {"title","text"}

Layout (little endian): magic b"MDRCORP1", int64 n, int64 blob_bytes, int64 title_off[n+1], int64 text_off[n+1],
uint8 intro[n], UTF-8 blob (titles then texts, addressed by the two offset tables).
"""
import json
import os
from collections.abc import Mapping

import numpy as np

MAGIC = b"MDRCORP1"


def _entry(v):
    if isinstance(v, (list, tuple)):  # [title, text, (is_intro)] form written by encode_corpus / used by the FEVER script
        return v[0], v[1], bool(v[2]) if len(v) > 2 else False
    return v["title"], v["text"], bool(v.get("intro", False))


def build_store(id2doc, store_path):
    """id2doc: path of the JSON file or the loaded dict; keys must be "0" .. str(n-1) (what encode_corpus writes)."""
    if isinstance(id2doc, str):
        with open(id2doc) as f:
            id2doc = json.load(f)
    n = len(id2doc)
    title_off = np.zeros(n + 1, np.int64)
    text_off = np.zeros(n + 1, np.int64)
    intro = np.zeros(n, np.uint8)
    titles, texts = [], []
    for i in range(n):
        try:
            t, x, it = _entry(id2doc[str(i)])
        except KeyError:
            raise ValueError(f"corpus dict keys must be '0'..'{n - 1}': '{i}' is missing") from None
        tb, xb = t.encode("utf-8"), x.encode("utf-8")
        titles.append(tb)
        texts.append(xb)
        title_off[i + 1] = title_off[i] + len(tb)
        text_off[i + 1] = text_off[i] + len(xb)
        intro[i] = it
    text_off += title_off[n]  # texts follow the titles in the blob
    tmp = store_path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([n, int(text_off[n])], np.int64).tobytes())
        f.write(title_off.tobytes())
        f.write(text_off.tobytes())
        f.write(intro.tobytes())
        for b in titles:
            f.write(b)
        for b in texts:
            f.write(b)
    os.replace(tmp, store_path)
    return store_path


class _Doc(dict):
    """{"title", "text"} (+ "intro"): a plain dict, so `json.dumps(id2doc[i])` and key access behave like the reference's."""


class CorpusStore(Mapping):
    def __init__(self, path):
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        if bytes(self._mm[:8]) != MAGIC:
            raise ValueError(f"{path} is not a corpus store (bad magic)")
        self.n, blob_bytes = (int(v) for v in self._mm[8:24].view(np.int64))
        o = 24
        self._title_off = self._mm[o:o + 8 * (self.n + 1)].view(np.int64)
        o += 8 * (self.n + 1)
        self._text_off = self._mm[o:o + 8 * (self.n + 1)].view(np.int64)
        o += 8 * (self.n + 1)
        self._intro = self._mm[o:o + self.n]
        o += self.n
        self._blob = self._mm[o:o + blob_bytes]
        if self._blob.shape[0] != blob_bytes:
            raise ValueError(f"{path} is truncated")
        self._with_intro = bool(self._intro.any())

    def __len__(self):
        return self.n

    def __iter__(self):
        return (str(i) for i in range(self.n))

    def _slice(self, off, i):
        return bytes(self._blob[int(off[i]):int(off[i + 1])]).decode("utf-8")

    def __getitem__(self, key):
        try:
            i = int(key)
        except (TypeError, ValueError):
            raise KeyError(key) from None
        if not 0 <= i < self.n or str(i) != str(key):
            raise KeyError(key)
        d = _Doc(title=self._slice(self._title_off, i), text=self._slice(self._text_off, i))
        if self._with_intro:
            d["intro"] = bool(self._intro[i])
        return d

    def as_list(self, key):
        """[title, text, is_intro] view of one entry (the FEVER script's corpus dict form)."""
        d = self[key]
        return [d["title"], d["text"], bool(self._intro[int(key)])]

"""Token arena: the corpus tokenised ONCE so that the hop-2 encoder inputs `<s> q </s></s> passage </s>` are
assembled on the device from the hop-1 ids (libmdrhip.so: mdr_assemble_hop2) instead of going
GPU -> host dict lookup -> Python tokenizer -> GPU between the hops as the reference does
(/root/reference/scripts/eval/eval_mhop_retrieval.py:158-169). Same result: RoBERTa pair encoding, HF
`longest_first` truncation to max_q_sp_len, empty passages replaced by their title with the hop-1 score set to -inf."""
import ctypes

import os

import numpy as np
import torch

from . import _lib


def arena_tag(tokenizer, roberta, max_tokens):
    """What a cached arena is valid for: the tokenisation RULE (2.11's prefix space, data.PREFIX_SPACE_2_11), the tokenizer class and
    vocabulary, the empty-text rule and the token cap. A cache written under another tag is rebuilt, never silently reused (ADVICE r3:
    an arena tokenised without the prefix space would disagree with the host path at the first BPE token of every passage)."""
    import hashlib
    import json

    from . import data as _data
    from .data import PREFIX_SPACE_2_11, is_roberta_family
    try:
        vocab = hashlib.sha256(json.dumps(sorted(tokenizer.get_vocab().items())).encode()).hexdigest()[:16]
    except Exception:
        vocab = "unknown"
    # (the class NAME without a trailing "Fast": the light loader takes it from tokenizer_config.json, AutoTokenizer of transformers 4 appends "Fast" for the same
    #  files, transformers 5 does not -- same BPE, same ids, one tag: ADVICE r5)
    name = tokenizer.__class__.__name__
    name = name[:-4] if name.endswith("Fast") else name
    rstrip = "|rstrip_segments_2_11=1" if (_data.RSTRIP_SEGMENTS_2_11 and is_roberta_family(tokenizer)) else ""  # (absent when off: caches written before the switch existed stay valid)
    return (f"arena-v2|prefix_space_2_11={int(_data.PREFIX_SPACE_2_11 and is_roberta_family(tokenizer))}{rstrip}|tokenizer={name}|vocab={vocab}"
            f"|empty_text_to_title={int(bool(roberta))}|max_tokens={max_tokens}")


def _npz_memmap(path):
    """{name: read-only np.memmap} over the members of an UNCOMPRESSED .npz (np.savez), or None when a member is compressed / not a plain array.
    np.load(npz) reads every member into fresh memory (3.6 GB for a 5 M-passage arena: 1.2 s + the pageable upload); a map costs nothing until the
    upload pipeline (mdr_upload_host) walks it."""
    import struct
    import zipfile
    out = {}
    try:
        with zipfile.ZipFile(path) as zf, open(path, "rb") as f:
            for info in zf.infolist():
                if info.compress_type != zipfile.ZIP_STORED or not info.filename.endswith(".npy"):
                    return None
                f.seek(info.header_offset)
                hdr = f.read(30)
                if hdr[:4] != b"PK\x03\x04":
                    return None
                n_name, n_extra = struct.unpack("<HH", hdr[26:30])
                start = info.header_offset + 30 + n_name + n_extra
                f.seek(start)
                version = np.lib.format.read_magic(f)
                shape, fortran, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
                if fortran or dtype.hasobject:
                    return None
                name = info.filename[:-4]
                if int(np.prod(shape)) == 0 or dtype.kind in "US":
                    f.seek(start)
                    out[name] = np.lib.format.read_array(f, allow_pickle=False)  # (the tag string, the empty `empty` array: tiny)
                else:
                    out[name] = np.memmap(path, dtype=dtype, mode="r", offset=f.tell(), shape=shape)
    except (OSError, ValueError, zipfile.BadZipFile, struct.error):
        return None
    return out


def _to_device(x, device, dtype):
    """torch tensor (any device) or numpy array / memmap -> contiguous device tensor of `dtype`; big host arrays go through libmdrhip's pinned
    double-buffer pipeline (mdr_upload_host) instead of a pageable `.to(device)`."""
    if torch.is_tensor(x):
        return x.to(device=device, dtype=dtype).contiguous()
    x = np.ascontiguousarray(x) if not isinstance(x, np.memmap) else x
    want = {torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}[dtype]
    if x.dtype != want:
        x = np.ascontiguousarray(x, dtype=want)
    device = torch.device(device)
    if device.type != "cuda":
        return torch.from_numpy(np.array(x)).to(device)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    out = torch.empty(x.shape, dtype=dtype, device=torch.device("cuda", dev_index))
    if x.size:
        _lib.check(_lib.lib().mdr_upload_host(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(x.ctypes.data), x.nbytes, dev_index,
                                              _lib.current_stream_ptr(out.device)))
    return out


class TokenArena:
    def __init__(self, tokens, offsets, empty=None, bos_id=0, eos_id=2, pad_id=1):
        """tokens / offsets / empty: torch tensors, or (from load(), lazily) numpy arrays / memmaps that only `.to(device)` turns into tensors."""
        lazy = not torch.is_tensor(tokens)
        self.tokens = tokens if lazy else tokens.to(dtype=torch.int32).contiguous()
        self.offsets = offsets if lazy else offsets.to(dtype=torch.int64).contiguous()
        self.empty = None if empty is None else (empty if lazy else empty.to(dtype=torch.uint8).contiguous())
        self.bos_id, self.eos_id, self.pad_id = bos_id, eos_id, pad_id
        self.n_docs = int(self.offsets.shape[0]) - 1

    def to(self, device):
        return TokenArena(_to_device(self.tokens, device, torch.int32), _to_device(self.offsets, device, torch.int64),
                          None if self.empty is None else _to_device(self.empty, device, torch.uint8), self.bos_id, self.eos_id, self.pad_id)

    # -- builders ----------------------------------------------------------------------------------------
    @classmethod
    def from_corpus(cls, id2doc, tokenizer, roberta=True, max_tokens=None):
        """id2doc: {"<row id>": {"title","text"}} (mhop.load_corpus_dict). Passages are tokenised WITHOUT special tokens, as the
        second segment of transformers 2.11's pair encoding sees them (data.prefix_space_2_11: a leading space for RoBERTa);
        an empty text falls back to the title and is flagged (eval_mhop_retrieval.py:162-165)."""
        from .data import is_roberta_family, prefix_space_2_11
        pre = prefix_space_2_11 if is_roberta_family(tokenizer) else (lambda t: t)
        n = len(id2doc)
        toks, offs, empty = [], np.zeros(n + 1, np.int64), np.zeros(n, np.uint8)
        step = 4096  # one tokenizer call per block of passages: the fast tokenizers encode a batch in parallel (5 M passages: minutes instead of an hour)
        for lo in range(0, n, step):
            texts = []
            for i in range(lo, min(n, lo + step)):
                doc = id2doc[str(i)]
                text = doc["text"]
                if roberta and text.strip() == "":
                    text = doc["title"]
                    empty[i] = 1
                texts.append(pre(text))
            for j, ids in enumerate(tokenizer(texts, add_special_tokens=False)["input_ids"]):
                if max_tokens is not None:
                    ids = ids[:max_tokens]  # never more than max_q_sp_len - 4 tokens can survive truncation
                toks.append(np.asarray(ids, np.int32))
                offs[lo + j + 1] = offs[lo + j] + len(ids)
        tokens = np.concatenate(toks) if toks else np.zeros(0, np.int32)
        return cls(torch.from_numpy(tokens), torch.from_numpy(offs), torch.from_numpy(empty))

    @classmethod
    def synthetic(cls, n_docs, device, seed=5, vocab=50265, min_len=60, max_len=300):
        """Passage lengths U[min_len, max_len], tokens uniform in [3, vocab) (bench / tests: no corpus text offline)."""
        g = torch.Generator(device=device).manual_seed(seed)
        lens = torch.randint(min_len, max_len + 1, (n_docs,), generator=g, device=device)
        offsets = torch.zeros(n_docs + 1, dtype=torch.int64, device=device)
        offsets[1:] = torch.cumsum(lens, 0)
        total = int(offsets[-1].item())
        tokens = torch.randint(3, vocab, (total,), generator=g, device=device, dtype=torch.int32)
        return cls(tokens, offsets, None)

    def save(self, path, tag=""):
        """np.savez appends ".npz" unless the name ends with it; `tag` (arena_tag) records what the tokens are valid for."""
        path = os.fspath(path)
        final = path if path.endswith(".npz") else path + ".npz"
        tmp = final + f".tmp{os.getpid()}.npz"  # written under another name and renamed: a rank polling for the file never reads a partial one
        def host(x):
            return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
        np.savez(tmp, tokens=host(self.tokens), offsets=host(self.offsets),
                 empty=(np.zeros(0, np.uint8) if self.empty is None else host(self.empty)), tag=np.array(str(tag)))
        os.replace(tmp, final)

    @classmethod
    def load(cls, path, expect_tag=None):
        """expect_tag: return None (caller rebuilds) when the file carries no tag or a different one."""
        z = _npz_memmap(path)  # members mapped, not read: `.to(device)` streams them through the upload pipeline
        if z is not None and {"tokens", "offsets", "empty"} <= set(z):
            if expect_tag is not None and ("tag" not in z or str(z["tag"]) != str(expect_tag)):
                return None
            return cls(z["tokens"], z["offsets"], z["empty"] if z["empty"].size else None)
        z = np.load(path)
        if expect_tag is not None and ("tag" not in z.files or str(z["tag"]) != str(expect_tag)):
            return None
        return cls(torch.from_numpy(z["tokens"]), torch.from_numpy(z["offsets"]), torch.from_numpy(z["empty"]) if z["empty"].size else None)

    # -- the device op -------------------------------------------------------------------------------------
    def assemble_hop2(self, q_ids, q_mask, doc_ids, hop1_scores=None, out_len=350):
        """q_ids/q_mask int64 cuda [B, Lq] (the question encoded as `<s> q </s>` right-padded), doc_ids int64 cuda [B, beam],
        hop1_scores float32 cuda [B, beam] or None (modified in place: -inf for empty passages).
        -> input_ids, attention_mask int64 cuda [B*beam, out_len]."""
        dev = q_ids.device
        B, Lq = q_ids.shape
        beam = doc_ids.shape[1]
        q_ids, q_mask, doc_ids = q_ids.contiguous(), q_mask.contiguous(), doc_ids.contiguous()
        if hop1_scores is not None and not hop1_scores.is_contiguous():
            raise ValueError("hop1_scores must be contiguous (it is updated in place)")
        ids = torch.empty((B * beam, out_len), dtype=torch.int64, device=dev)
        mask = torch.empty_like(ids)
        vp = ctypes.c_void_p
        _lib.check(_lib.lib().mdr_assemble_hop2(vp(q_ids.data_ptr()), vp(q_mask.data_ptr()), B, Lq, vp(doc_ids.data_ptr()), beam, vp(self.tokens.data_ptr()),
                                                vp(self.offsets.data_ptr()), vp(self.empty.data_ptr() if self.empty is not None else None), self.n_docs,
                                                vp(hop1_scores.data_ptr() if hop1_scores is not None else None), out_len, self.bos_id, self.eos_id,
                                                self.pad_id, vp(ids.data_ptr()), vp(mask.data_ptr()), _lib.current_stream_ptr(dev)))
        return ids, mask

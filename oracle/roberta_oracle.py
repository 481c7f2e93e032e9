"""oracle/roberta_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the encoder forward the reference runs on the hot path:

    RobertaRetriever.encode_q -> encode_seq    /root/reference/mdr/retrieval/models/mhop_retriever.py:40-41,23-26
        cls = self.encoder(input_ids, mask)[0][:, 0, :]          (:24)
        return self.project(cls)   # Linear(768,768)+LayerNorm   (:21,25)
    RobertaCtxEncoder.forward                   /root/reference/mdr/retrieval/models/retriever.py:186-190
        (same arithmetic, keys input_ids / input_mask -> {'embed': ...})

`self.encoder` is HuggingFace `RobertaModel` (transformers==2.11.0 pinned at
/root/reference/requirements.txt:1 -- third party, NOT in /root/reference). Its published forward
is restated here: position ids from `input_ids != pad_id` (cumsum * mask + pad_id),
word + position + token-type(0) embeddings -> LayerNorm; 12 post-LN layers of
softmax(Q K^T / sqrt(64) + additive mask) V -> dense + residual + LN -> dense + erf-GELU -> dense +
residual + LN; the pooler is never evaluated because `[0]` (sequence output) is taken.

Pinned by tests/golden/encoder_*.npz: outputs of the reference classes themselves, imported in the
build container from /root/reference by oracle/gen_golden.py (eager attention, fp32, CPU).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

try:  # scipy is present in the image; keep a fallback so the oracle has no hard dependency
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gelu(x):
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def position_ids(input_ids, pad_id):
    m = (input_ids != pad_id).astype(np.int64)
    return np.cumsum(m, axis=1) * m + pad_id


def encode(sd, geom, input_ids, attention_mask, dtype=np.float32):
    """-> [B, hidden] embedding (project(CLS)), arithmetic in `dtype` (float32 or float64)."""
    W = lambda k: np.asarray(sd[k]).astype(dtype)
    H, nh, eps = geom["hidden"], geom["heads"], geom["ln_eps"]
    hd = H // nh
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask)
    B, L = ids.shape
    pos = position_ids(ids, geom["pad_id"])
    x = (W("encoder.embeddings.word_embeddings.weight")[ids]
         + W("encoder.embeddings.position_embeddings.weight")[pos]
         + W("encoder.embeddings.token_type_embeddings.weight")[0])
    x = layer_norm(x, W("encoder.embeddings.LayerNorm.weight"), W("encoder.embeddings.LayerNorm.bias"), eps)
    # additive mask: transformers 2.11 uses (1-mask)*-10000; masked probabilities underflow to 0 either way
    add_mask = ((1.0 - mask.astype(dtype)) * dtype(-10000.0))[:, None, None, :]
    for i in range(geom["layers"]):
        p = f"encoder.encoder.layer.{i}."
        lin = lambda t, n: t @ W(p + n + ".weight").T + W(p + n + ".bias")
        q = lin(x, "attention.self.query").reshape(B, L, nh, hd).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(B, L, nh, hd).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(B, L, nh, hd).transpose(0, 2, 1, 3)
        s = q @ k.transpose(0, 1, 3, 2) / dtype(math.sqrt(hd)) + add_mask
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        pr = e / e.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, L, H)
        x = layer_norm(lin(ctx, "attention.output.dense") + x,
                       W(p + "attention.output.LayerNorm.weight"), W(p + "attention.output.LayerNorm.bias"), eps)
        h = gelu(lin(x, "intermediate.dense"))
        x = layer_norm(lin(h, "output.dense") + x,
                       W(p + "output.LayerNorm.weight"), W(p + "output.LayerNorm.bias"), eps)
    cls = x[:, 0, :]
    y = cls @ W("project.0.weight").T + W("project.0.bias")
    return layer_norm(y, W("project.1.weight"), W("project.1.bias"), eps).astype(dtype)

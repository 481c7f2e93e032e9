"""oracle/seeded.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Framework-independent, counter-based tensor generator (SURVEY.md §8c "weights-from-seed rule").
Fixtures under tests/golden/ never store encoder weights: both oracle/gen_golden.py (which feeds
them to the imported reference model) and the tests (which feed them to the HIP path) derive every
tensor of the q_encoder.pt schema from (seed, tensor name, shape).

splitmix64(counter) -> two uniforms -> Box-Muller normal. Pure numpy uint64 arithmetic, so the
values are bit-reproducible on any machine.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, name: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in (0,1), keyed by (seed, name, stream)."""
    key = (_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15) ^ (stream * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + np.uint64(key)) & _M64
    bits = _splitmix64(ctr)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    u1 = uniform01(seed, name, n, 0)
    u2 = uniform01(seed, name, n, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def integers(seed: int, name: str, shape, lo: int, hi: int) -> np.ndarray:
    """int64 in [lo, hi)."""
    n = int(np.prod(shape))
    u = uniform01(seed, name, n, 2)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).clip(lo, hi - 1).reshape(shape)


# ---------------------------------------------------------------------------------------------
# q_encoder.pt schema (key names are those of
# /root/reference/mdr/retrieval/models/mhop_retriever.py:20-21 wrapping HF RobertaModel)
# ---------------------------------------------------------------------------------------------
ROBERTA_BASE = dict(vocab=50265, hidden=768, layers=12, heads=12, ffn=3072, max_pos=514, ln_eps=1e-5, pad_id=1)
TINY = dict(vocab=512, hidden=128, layers=2, heads=2, ffn=256, max_pos=514, ln_eps=1e-5, pad_id=1)
# roberta-base WIDTH (the kernels' real tile shapes) at depth 2 = one full layer + the CLS-only last layer. fp16 operand rounding is a
# discontinuous map: two correct implementations of the SAME apex-O1 arithmetic that differ by fp32 summation order decorrelate
# a little more at every rounding (at depth 12 their distance is ~0.7 of the regime's own error against fp32), so the
# sharp operand-rounded parity bar is only meaningful on a SHALLOW stack -- this one.
WIDE2 = dict(vocab=50265, hidden=768, layers=2, heads=12, ffn=3072, max_pos=514, ln_eps=1e-5, pad_id=1)


def state_dict_shapes(geom, with_pooler=True):
    H, F = geom["hidden"], geom["ffn"]
    sd = {
        "encoder.embeddings.word_embeddings.weight": (geom["vocab"], H),
        "encoder.embeddings.position_embeddings.weight": (geom["max_pos"], H),
        "encoder.embeddings.token_type_embeddings.weight": (1, H),
        "encoder.embeddings.LayerNorm.weight": (H,),
        "encoder.embeddings.LayerNorm.bias": (H,),
    }
    for i in range(geom["layers"]):
        p = f"encoder.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            sd[p + f"attention.self.{n}.weight"] = (H, H)
            sd[p + f"attention.self.{n}.bias"] = (H,)
        sd[p + "attention.output.dense.weight"] = (H, H)
        sd[p + "attention.output.dense.bias"] = (H,)
        sd[p + "attention.output.LayerNorm.weight"] = (H,)
        sd[p + "attention.output.LayerNorm.bias"] = (H,)
        sd[p + "intermediate.dense.weight"] = (F, H)
        sd[p + "intermediate.dense.bias"] = (F,)
        sd[p + "output.dense.weight"] = (H, F)
        sd[p + "output.dense.bias"] = (H,)
        sd[p + "output.LayerNorm.weight"] = (H,)
        sd[p + "output.LayerNorm.bias"] = (H,)
    if with_pooler:
        sd["encoder.pooler.dense.weight"] = (H, H)
        sd["encoder.pooler.dense.bias"] = (H,)
    sd["project.0.weight"] = (H, H)
    sd["project.0.bias"] = (H,)
    sd["project.1.weight"] = (H,)
    sd["project.1.bias"] = (H,)
    return sd


def make_state_dict(seed: int, geom, with_pooler=True):
    """name -> float32 ndarray. Linear/embedding weights N(0, std), LN gamma 1+0.1n, beta/bias 0.1n
    (non-trivial LN and bias so a dropped bias or a swapped gamma/beta is caught).

    std is 0.02 for embeddings and 1.5/sqrt(fan_in) for Linear weights: with the HF default 0.02
    every layer's contribution to the residual stream is ~1e-2 of the LayerNorm'd input, and a
    12-layer model's CLS output would barely depend on layers 1..11 -- a wrong attention kernel
    would pass. The larger std keeps each sub-layer's output O(1)."""
    out = {}
    for name, shape in state_dict_shapes(geom, with_pooler).items():
        if name.endswith("LayerNorm.weight") or name == "project.1.weight":
            out[name] = normal(seed, name, shape, 0.1, 1.0)
        elif name.endswith("bias"):
            out[name] = normal(seed, name, shape, 0.1, 0.0)
        elif "embeddings" in name:
            out[name] = normal(seed, name, shape, 0.5, 0.0)
        else:
            out[name] = normal(seed, name, shape, 1.5 / np.sqrt(shape[1]), 0.0)
    return out


def make_token_batch(seed: int, name: str, B: int, L: int, vocab: int, min_len: int = 1, pad_id: int = 1,
                     pad_fill: int = 1):
    """ids int64 [B,L] = <s> tok... </s> pad..., mask int64 [B,L]; row 0 is always full length,
    row 1 (if any) has the minimum length."""
    lens = integers(seed, name + ".len", (B,), max(min_len, 2), L + 1)
    lens[0] = L
    if B > 1:
        lens[1] = max(min_len, 2) if min_len >= 2 else max(min_len, 1)
    body = integers(seed, name + ".tok", (B, L), 3, vocab)
    ids = np.full((B, L), pad_fill, np.int64)
    mask = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = body[b, :n]
        ids[b, 0] = 0
        if n >= 2:
            ids[b, n - 1] = 2
        mask[b, :n] = 1
    return ids, mask

"""oracle/roberta_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The arithmetic of oracle/roberta_oracle.py (the numpy restatement of RobertaRetriever.encode_q,
/root/reference/mdr/retrieval/models/mhop_retriever.py:23-26,40-41 + HF RobertaModel 2.11) written with torch ops, so
that the fp64 / fp32 reference embeddings of THOUSANDS of sequences of roberta-base geometry can be produced on the GPU
box in seconds (the numpy version needs minutes there). Plain torch fp64 -- no custom kernels, nothing shared with the
product path. Pinned by tests/test_oracle_encoder.py::test_torch_restatement_equals_numpy_restatement (<= 1e-9 in fp64),
the numpy restatement in turn by the reference's own outputs (tests/golden/encoder_*.npz).

`o1`: apex-O1 numerics restated (the regime of eval_mhop_retrieval.py:86-90; pinned by the fixtures `*.embed_o1ops` / `*.embed_o1lit`
that oracle/gen_golden.py produces from the IMPORTED reference model under a torch-function mode):
    "operands"  Linear / matmul inputs and weights rounded to fp16, products and sums in `dtype`, outputs not rounded
    "literal"   additionally the outputs of Linear (matmul result, then the bias add) and of the attention matmuls rounded to fp16
    "sums16"    "operands" + the outputs of the two Linears in front of the LayerNorms (attention.output.dense, output.dense) rounded to fp16 once,
                after the bias add, BEFORE the residual add -- the dataflow of mdr_encoder_config.residual_fp32 = 2 (fp32 residual stream; apex O1's
                F.linear returns fp16 there). Restated, not produced by the imported model: between the two fixture regimes by construction.

Only tests/ (and oracle/gen_golden.py) may import this module.
"""
import math

import torch


def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def encode(sd, geom, input_ids, attention_mask, dtype=torch.float64, device="cpu", chunk=64, o1=None):
    """sd: name -> numpy / torch tensors (fp32 checkpoint values). -> [B, hidden] tensor of `dtype` on `device`."""
    assert o1 in (None, "operands", "literal", "sums16")
    W = {k: torch.as_tensor(v).to(device=device, dtype=dtype) for k, v in sd.items() if "pooler" not in k}

    def r(t):  # fp16 rounding of a GEMM operand (or, literal mode, of a GEMM output)
        return t.to(torch.float16).to(dtype) if o1 else t

    def ro(t):
        return t.to(torch.float16).to(dtype) if o1 == "literal" else t

    def linear(t, w, b):
        y = ro(r(t) @ r(w).T)
        return ro(y + r(b)) if o1 == "literal" else y + b
    H, nh, eps, pad = geom["hidden"], geom["heads"], geom["ln_eps"], geom["pad_id"]
    hd = H // nh
    ids_all = torch.as_tensor(input_ids).to(device)
    mask_all = torch.as_tensor(attention_mask).to(device)
    outs = []
    for lo in range(0, ids_all.shape[0], chunk):
        ids, mask = ids_all[lo:lo + chunk], mask_all[lo:lo + chunk]
        B, L = ids.shape
        m = (ids != pad).long()
        pos = torch.cumsum(m, 1) * m + pad
        x = (W["encoder.embeddings.word_embeddings.weight"][ids] + W["encoder.embeddings.position_embeddings.weight"][pos]
             + W["encoder.embeddings.token_type_embeddings.weight"][0])
        x = layer_norm(x, W["encoder.embeddings.LayerNorm.weight"], W["encoder.embeddings.LayerNorm.bias"], eps)
        add_mask = ((1.0 - mask.to(dtype)) * -10000.0)[:, None, None, :]
        for i in range(geom["layers"]):
            p = f"encoder.encoder.layer.{i}."

            def lin(t, n):
                return linear(t, W[p + n + ".weight"], W[p + n + ".bias"])
            q = lin(x, "attention.self.query").reshape(B, L, nh, hd).permute(0, 2, 1, 3)
            k = lin(x, "attention.self.key").reshape(B, L, nh, hd).permute(0, 2, 1, 3)
            v = lin(x, "attention.self.value").reshape(B, L, nh, hd).permute(0, 2, 1, 3)
            s = ro(r(q) @ r(k).transpose(-1, -2)) / math.sqrt(hd) + add_mask
            pr = torch.softmax(s, -1)
            ctx = ro(r(pr) @ r(v)).permute(0, 2, 1, 3).reshape(B, L, H)
            def pre_ln(t):  # the Linear output that meets the fp32 residual
                return t.to(torch.float16).to(dtype) if o1 == "sums16" else t
            x = layer_norm(pre_ln(lin(ctx, "attention.output.dense")) + x, W[p + "attention.output.LayerNorm.weight"],
                           W[p + "attention.output.LayerNorm.bias"], eps)
            h = lin(x, "intermediate.dense")
            h = 0.5 * h * (1.0 + torch.erf(h / math.sqrt(2.0)))
            x = layer_norm(pre_ln(lin(h, "output.dense")) + x, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)
        y = linear(x[:, 0, :], W["project.0.weight"], W["project.0.bias"])
        outs.append(layer_norm(y, W["project.1.weight"], W["project.1.bias"], eps))
    return torch.cat(outs)

#!/usr/bin/env python3
"""A/B harness for encoder GEMM kernels: one process per MDR_GEMM_CFG value runs the same forward and saves the
embeddings + the HIP-event time; the parent compares outputs (same K order => expected bit-identical) and prints times.
usage: python scripts/measure/gpu_gemm_ab.py [B] [L] [cfgA] [cfgB]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(B, L, out):
    import torch
    sys.path.insert(0, ROOT)
    from multihop_dense_retrieval_amd.retriever import RobertaRetriever
    dev = torch.device("cuda", 0)
    m = RobertaRetriever.random_init(device=dev, seed=3)
    g = torch.Generator(device=dev).manual_seed(5)
    lens = torch.randint(L // 2, L + 1, (B,), generator=g, device=dev)
    ids = torch.randint(3, 50265, (B, L), generator=g, device=dev)
    pos = torch.arange(L, device=dev)[None, :]
    mask = (pos < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.ones_like(ids))
    ids[:, 0] = 0
    for _ in range(3):
        e = m.encode_q(ids, mask, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        e = m.encode_q(ids, mask, None)
    e1.record()
    torch.cuda.synchronize()
    torch.save({"e": e.cpu(), "ms": e0.elapsed_time(e1) / 10, "tokens": int(mask.sum())}, out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    cfgs = sys.argv[3:5] if len(sys.argv) > 4 else ["0", "5"]
    res = {}
    for c in cfgs:
        env = dict(os.environ, MDR_GEMM_CFG=c)
        out = f"/tmp/gemm_ab_{c}.pt"
        r = subprocess.run([sys.executable, __file__, "--child", str(B), str(L), out], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(f"cfg {c} FAILED rc={r.returncode}\n{r.stderr[-2000:]}")
            sys.exit(1)
        res[c] = torch.load(out)
        print(f"cfg {c}: B={B} L={L} tokens={res[c]['tokens']} forward {res[c]['ms']:.3f} ms  finite={bool(torch.isfinite(res[c]['e']).all())}")
    a, b = res[cfgs[0]]["e"], res[cfgs[1]]["e"]
    d = (a - b).abs().max().item()
    print(f"max |cfg{cfgs[0]} - cfg{cfgs[1]}| = {d:.3e}  (rows differing: {int(((a - b).abs().amax(1) > 0).sum())} of {a.shape[0]})")
    sys.exit(0 if d <= 1e-3 else 2)

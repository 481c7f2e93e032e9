"""Attention-kernel check of a VARIANT library build against the product library: one encoder forward per case in a fresh process per library (MDR_LIB_PATH),
outputs compared here.   python scripts/measure/gpu_attn_check.py libmdrhip_variant.so [ncases]      (cases: B, min len, max len, padded L)"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(os.path.dirname(HERE)), "multihop_dense_retrieval_amd")
CASES = [(2, 150, 150, 200), (2, 40, 60, 200), (3, 129, 256, 260), (4, 257, 344, 350), (40, 20, 340, 350), (100, 72, 344, 350)]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    B, lo, hi, L, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from multihop_dense_retrieval_amd.retriever import RobertaRetriever
    torch.manual_seed(0)
    m = RobertaRetriever.random_init(device="cuda", seed=3)
    lens = torch.randint(lo, hi + 1, (B,))
    ids = torch.ones((B, L), dtype=torch.int64)
    mask = torch.zeros((B, L), dtype=torch.int64)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = torch.randint(3, 50000, (n,))
        ids[b, 0], ids[b, n - 1] = 0, 2
        mask[b, :n] = 1
    y = m.encode_q(ids.cuda(), mask.cuda(), None)
    torch.cuda.synchronize()
    np.save(out, y.float().cpu().numpy())
    sys.exit(0)

new = sys.argv[1]
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else len(CASES)
for case in CASES[:ncases]:
    outs = []
    for lib in ("libmdrhip.so", new):
        out = f"/tmp/attn_check_{lib}.npy"
        if os.path.exists(out):
            os.remove(out)
        env = dict(os.environ, MDR_LIB_PATH=os.path.join(PKG, lib))
        try:
            r = subprocess.run([sys.executable, __file__, "--child", *map(str, case), out], env=env, timeout=60, capture_output=True, text=True)
            tail = (r.stderr or "").strip().splitlines()[-1:] if r.returncode else []
            outs.append(np.load(out) if r.returncode == 0 else f"rc={r.returncode} {tail}")
        except subprocess.TimeoutExpired:
            outs.append("TIMEOUT")
    if all(isinstance(o, np.ndarray) for o in outs):
        print(f"case B={case[0]} len {case[1]}..{case[2]} L={case[3]}: max |diff| {np.abs(outs[0] - outs[1]).max():.3e}  finite {np.isfinite(outs[1]).all()}", flush=True)
    else:
        print(f"case {case}: product={'ok' if isinstance(outs[0], np.ndarray) else outs[0]} variant={'ok' if isinstance(outs[1], np.ndarray) else outs[1]}", flush=True)
        if not isinstance(outs[1], np.ndarray):
            break  # the GPU may be in a bad state after a fault: stop here

"""Per-workgroup timeline of the attention kernel for L > 128 (a -DMDR_ATTN_ABL=9 build selected with MDR_LIB_PATH: the ring kernel; with -DMDR_ATTN_RING=0 the streaming kernel): when does each workgroup wait for its
K/V, when does it compute, and what runs beside it on the same CU? Hop-2-shaped forward as scripts/measure/gpu_enc_forward.py; the stamps are those of the
LAST streaming-attention launch of the forward (layer 11).

    python -m multihop_dense_retrieval_amd.build -DMDR_ATTN_ABL=9 --out=libmdrhip_attn_timeline.so
    MDR_LIB_PATH=$PWD/multihop_dense_retrieval_amd/libmdrhip_attn_timeline.so python scripts/measure/gpu_attn_timeline.py
"""
import collections
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd import _lib  # noqa: E402
from multihop_dense_retrieval_amd.retriever import RobertaRetriever  # noqa: E402

lo, hi = (int(v) for v in os.environ.get("ENC_LEN", "72,344").split(","))
B, L, HEADS = 100, int(os.environ.get("ENC_L", "350")), 12
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
ids, mask = ids.cuda(), mask.cuda()
for _ in range(4):
    m.encode_q(ids, mask, None)
torch.cuda.synchronize()
nz = (L + 127) // 128
nwg = HEADS * B * nz
buf = (ctypes.c_uint64 * (nwg * 8))()
_lib.check(_lib.lib().mdr_test_attn_stamps(buf, nwg))
s = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
real = s[:, 4] > 0
r = s[real]
t0 = r[:, 0].min()
T = (r[:, :4] - t0) * 10.0 / 1e3  # us (100 MHz ticks)
print(f"workgroups launched {nwg}, with work {int(real.sum())}; kernel span {T[:, 3].max():.1f} us (first entry to last exit), tokens {int(lens.sum())}")
ld, c0, rest, life = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 3] - T[:, 0]
for name, v in (("entry -> K/V landed", ld), ("first query block (S, softmax, PV, stores)", c0), ("rest (second block / further chunks)", rest), ("lifetime", life)):
    print(f"  {name:46s} mean {v.mean():6.2f} us  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}")
if r[:, 7].max() > 0:  # the ring kernel: ticks waited at the tops of jobs 1.. (its first wait is "entry -> K/V landed": Q and job 0)
    tw = r[:, 7] / 100.0
    print(f"  waited at later job tops (wait + barrier)        mean {tw.mean():6.2f} us  p50 {np.percentile(tw, 50):6.2f}  p90 {np.percentile(tw, 90):6.2f}   ({tw.sum() / life.sum():.2f} of the lifetimes)")
# per CU: (xcc, se, cu) from XCC_ID[3:0], HW_ID se_id[15:13], sh_id[12], cu_id[11:8]
cu_key = (r[:, 6] & 15) * 4096 + ((r[:, 5] >> 8) & 0xFF)
per_cu = collections.defaultdict(list)
for i in range(len(r)):
    per_cu[int(cu_key[i])].append(i)
print(f"  distinct CUs seen {len(per_cu)}; workgroups per CU: mean {np.mean([len(v) for v in per_cu.values()]):.2f} max {max(len(v) for v in per_cu.values())}")
# overlap on a CU: of the time a workgroup waits for its loads, how much lies inside another resident workgroup's compute (K/V landed .. exit)?
wait_tot = wait_beside_compute = wait_beside_wait = 0.0
comp_tot = comp_beside_comp = 0.0
for idx in per_cu.values():
    for i in idx:
        a0, a1, a3 = T[i, 0], T[i, 1], T[i, 3]
        wait_tot += a1 - a0
        comp_tot += a3 - a1
        for j in idx:
            if j == i:
                continue
            b0, b1, b3 = T[j, 0], T[j, 1], T[j, 3]
            wait_beside_compute += max(0.0, min(a1, b3) - max(a0, b1))
            wait_beside_wait += max(0.0, min(a1, b1) - max(a0, b0))
            comp_beside_comp += max(0.0, min(a3, b3) - max(a1, b1))
print(f"  of the load-wait time, {wait_beside_compute / wait_tot:.2f} lies beside another workgroup's compute on the same CU, {wait_beside_wait / wait_tot:.2f} beside another's load wait")
print(f"  of the compute time, {comp_beside_comp / comp_tot:.2f} lies beside another workgroup's compute on the same CU")
# how much slower is a workgroup's compute when the other slot of its CU computes too? (items of similar work: merged sequences of 180..230 tokens)
rows = []
for idx in per_cu.values():
    for i in idx:
        if not (180 <= r[i, 4] <= 230):
            continue
        a1, a3 = T[i, 1], T[i, 3]
        ov = sum(max(0.0, min(a3, T[j, 3]) - max(a1, T[j, 1])) for j in idx if j != i)
        rows.append((ov / max(a3 - a1, 1e-9), a3 - a1))
rows = np.array(rows)
if len(rows):
    for lo_, hi_ in ((0.0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.01)):
        sel = (rows[:, 0] >= lo_) & (rows[:, 0] < hi_)
        if sel.any():
            print(f"  merged 180..230 tokens, share of compute beside the other slot's compute in [{lo_:.2f}, {hi_:.2f}): n {int(sel.sum()):4d}  compute {rows[sel, 1].mean():6.2f} us")
# per CU: share of the kernel span with 0 / 1 / 2 workgroups in their compute phase
span_all = T[:, 3].max()
occ = np.zeros(4)
for idx in per_cu.values():
    ev = sorted([(T[i, 1], 1) for i in idx] + [(T[i, 3], -1) for i in idx])
    t_prev, n = 0.0, 0
    for t_, d_ in ev:
        occ[min(n, 3)] += t_ - t_prev
        t_prev, n = t_, n + d_
    occ[min(n, 3)] += span_all - t_prev
occ /= occ.sum()
print(f"  CU time with 0 / 1 / 2 / 3+ workgroups computing: {occ[0]:.2f} / {occ[1]:.2f} / {occ[2]:.2f} / {occ[3]:.2f}")
# chip level: how many workgroups are waiting for loads / computing in each 2-us bin
print("  t(us)  waiting  computing  (of 512 slots)")
for b0 in np.arange(0.0, span_all + 2.0, 2.0)[:-1]:
    w = ((T[:, 0] < b0 + 2) & (T[:, 1] > b0)).sum()
    c = ((T[:, 1] < b0 + 2) & (T[:, 3] > b0)).sum()
    print(f"  {b0:5.0f}  {int(w):7d}  {int(c):9d}")
ln = r[:, 4]
for name, sel in (("len <= 128", ln <= 128), ("129..256 (merged)", (ln > 128) & (ln <= 256)), ("> 256 (two chunks, one block per workgroup)", ln > 256)):
    if sel.any():
        print(f"  {name:46s} n {int(sel.sum()):5d}  wait {ld[sel].mean():6.2f}  first block {c0[sel].mean():6.2f}  rest {rest[sel].mean():6.2f}  lifetime {life[sel].mean():6.2f} us")

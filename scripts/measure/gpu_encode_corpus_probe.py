"""Where does encode_shard() spend its time on the GPU box? (loader alone; the predict loop with per-phase host timers)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.utils.data import DataLoader
from multihop_dense_retrieval_amd import encode_corpus
from multihop_dense_retrieval_amd.retriever import RobertaCtxEncoder, move_to_cuda
n, bs, Lmax = 100000, 1000, 300
g = torch.Generator().manual_seed(7)
lens = torch.randint(20, Lmax + 1, (n,), generator=g)
offs = torch.zeros(n + 1, dtype=torch.int64); offs[1:] = torch.cumsum(lens, 0)
toks = torch.randint(3, 50265, (int(offs[-1]),), generator=g)
class Synth(torch.utils.data.Dataset):
    def __len__(self): return n
    def __getitem__(self, i):
        ids = toks[offs[i]:offs[i + 1]].view(1, -1)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
ds = Synth()
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
def mk(nw, pin):
    return DataLoader(encode_corpus._Indexed(ds, 0, n), batch_size=16 * bs, collate_fn=encode_corpus.LengthBucketCollate(bs), num_workers=nw, pin_memory=pin)
for nw, pin in ((8, False), (8, True), (16, True)):
    t = time.perf_counter(); c = 0
    for w in mk(nw, pin):
        for rows, b in w: c += rows.numel()
    el = time.perf_counter() - t
    print(f"loader alone workers {nw} pin {pin}: {c / el:.0f} passages/s ({el:.2f} s)")
model = RobertaCtxEncoder.random_init(device=torch.device("cuda", 0), seed=3)
out = np.zeros((n, 768), np.float32)
stager = encode_corpus.DeviceStager(torch.device("cuda", 0))
for rep in range(3):
    T = dict(next=0.0, gpuwait=0.0, h2d=0.0, fwd=0.0, d2h=0.0, flush=0.0)
    t_all = time.perf_counter()
    it = iter(mk(8, False)); pending = None
    while True:
        t = time.perf_counter()
        try: window = next(it)
        except StopIteration: break
        T["next"] += time.perf_counter() - t
        for rows, batch in window:
            t = time.perf_counter(); torch.cuda.synchronize(); T["gpuwait"] += time.perf_counter() - t
            t = time.perf_counter(); b = encode_corpus.expand_compact(batch, stager); T["h2d"] += time.perf_counter() - t
            t = time.perf_counter()
            with torch.no_grad(): e = model(b)["embed"]
            T["fwd"] += time.perf_counter() - t
            t = time.perf_counter()
            host = hostbuf[len(T) and (id(pending) & 0) or 0] if False else torch.empty(e.shape, dtype=e.dtype, pin_memory=True); host.copy_(e, non_blocking=True); ev = torch.cuda.Event(); ev.record()
            T["d2h"] += time.perf_counter() - t
            t = time.perf_counter()
            if pending is not None:
                pending[2].synchronize(); out[pending[0].numpy()] = pending[1].numpy()
            pending = (rows, host, ev)
            T["flush"] += time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"rep {rep}: total {time.perf_counter() - t_all:.2f} s; host time by phase:", {k: round(v, 2) for k, v in T.items()}, "captures", model.graph_captures, "replays", model.graph_replays)

#!/usr/bin/env python3
"""Ablation of the CLI's host pipeline on the real encoder + index (1 M rows): which part of a batch's 13 ms is not GPU time?"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cli_bench_assets  # noqa: E402
from multihop_dense_retrieval_amd import eval_mhop_retrieval as E, mhop  # noqa: E402
from multihop_dense_retrieval_amd import pipeline as P  # noqa: E402
from multihop_dense_retrieval_amd.arena import TokenArena, arena_tag  # noqa: E402
from multihop_dense_retrieval_amd.retriever import RobertaRetriever, load_saved  # noqa: E402

rows, nq = int(os.environ.get("ROWS", "1000000")), int(os.environ.get("NQ", "4000"))
import transformers  # noqa: E402
tok0 = cli_bench_assets.make_tokenizer()
pool = P.TokenizerPool(tok0, int(os.environ.get("WORKERS", "12")))  # forked before the device is touched
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
out_dir = "/dev/shm/mdr_cli_ablate"
a = cli_bench_assets.build(out_dir, rows, nq, dev, log=lambda m: None)
tok = transformers.AutoTokenizer.from_pretrained(a["model_name"])
cfg = E._load_config(a["model_name"])
model = load_saved(RobertaRetriever(cfg, None), a["model_path"], exact=False)
model.to(dev).eval()
model.capture_on_first_use = True
index = E.load_index(a["indexpath"], d=768)
id2doc = mhop.load_corpus_dict(a["corpus_dict"])
arena = TokenArena.load(a["corpus_dict"] + ".arena.npz", expect_tag=arena_tag(tok, True, 350)).to(dev)
items = [json.loads(l) for l in open(a["raw_data"])]
qs = [mhop.strip_question(it["question"]) for it in items]


def finish(ann, D, I, D2, I2):
    chains = mhop.rank_paths(D, I, D2, I2, 1, 1)
    ms, recs = [], []
    for an, ch in zip(ann, chains):
        m = mhop.question_metrics(ch, an["sp"], id2doc)
        ms.append(m)
        recs.append(mhop.output_record(an, ch, id2doc))
    return ms, recs


def run(name, fuse=True, use_arena=True, finish_fn=finish, patch=None, depth=None, fw=int(os.environ.get("FW", "3"))):
    pipe = P.TwoHopPipeline(model, index, pool, id2doc, finish_fn, batch_size=100, beam=1, max_q_len=70, max_q_sp_len=350, arena=arena if use_arena else None,
                            device=dev, depth=depth, fuse=fuse, finish_workers=fw)
    if patch:
        patch(pipe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(qs, items)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    pipe.close()
    nb = -(-len(qs) // 100)
    print(f"{name:42s} {el / nb * 1e3:7.3f} ms/batch  gpu_stage {pipe.stats['gpu_stage_ms']}  main_s { {k: round(v, 3) for k, v in pipe.stats['main_s'].items()} } "
          f"wait1 {pipe.stats['wait_tok1_s']:.3f} wait2 {pipe.stats['wait_tok2_s']:.3f} finish {pipe.stats['finish_busy_s']:.3f}", flush=True)


run("warm (captures)")
if os.environ.get("ONLY"):
    for nm in os.environ["ONLY"].split(","):
        if nm == "noop":
            run("device+fused, finish = noop", finish_fn=lambda *a_: ([], []))
        if nm == "fused":
            run("device+fused")
        if nm == "plain":
            run("device, not fused", fuse=False)
        if nm == "host":
            run("host path (default flags), depth 8", fuse=False, use_arena=False)
            run("host path, fused, depth 8", fuse=True, use_arena=False)
    pool.close()
    import shutil
    shutil.rmtree(out_dir, ignore_errors=True)
    sys.exit(0)
run("device+fused")
run("device, not fused", fuse=False)
run("device+fused, finish = noop", finish_fn=lambda *a_: ([], []))


def no_d2h(pipe):
    z = {}
    def d2h(t):
        k = (tuple(t.shape), t.dtype)
        if k not in z:
            z[k] = torch.zeros(t.shape, dtype=t.dtype)
        return z[k]
    pipe._d2h = d2h
    pipe._release = lambda *h: None
run("device+fused, no D2H copies", patch=no_d2h)


def cached_h2d(pipe):
    cache = {}
    def h2d(arr):
        k = (arr.shape, str(arr.dtype))
        if k not in cache:
            cache[k] = torch.from_numpy(arr).to(dev)
        return cache[k]
    pipe._h2d = h2d
run("device+fused, no H2D copies (cached)", patch=cached_h2d)


def both(pipe):
    no_d2h(pipe)
    cached_h2d(pipe)
run("device+fused, no H2D, no D2H", patch=both)
run("device+fused, no H2D/D2H, finish noop", patch=both, finish_fn=lambda *a_: ([], []))
os.environ["X"] = "1"
old = sys.getswitchinterval()
run("host path (default flags), depth 8", fuse=False, use_arena=False)
run("host path, fused, depth 8", fuse=True, use_arena=False)
pool.close()
import shutil
shutil.rmtree(out_dir, ignore_errors=True)

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native MDR hot path.

    python bench.py --gpus N --steps K --warmup W

metric  : queries/sec of the 2-hop beam-search retrieval loop (BASELINE.json) over a synthetic
          5M x 768 fp32 index, 100-question batches, beam=1 topk=1 (BASELINE configs[2] shape).
step    : one batch of 100 questions through hop-1 encode -> MIPS -> hop-2 encode -> MIPS -> path rank
          (scripts/eval/eval_mhop_retrieval.py:142-206 of the reference), inputs resident in HBM.
N > 1   : launched by torch.distributed.run, one rank per GPU. The 5M-row corpus is row-sharded over the ranks (the
          north star's layout). Default --scaling weak: every rank owns its own batch of 100 questions (global batch
          100*N): it encodes them, ONE RCCL all_gather shares the embeddings, every rank searches all 100*N queries in
          its shard, ONE all_gather per hop exchanges the per-shard top-k lists, every rank merges and continues with its
          own questions. Per-GPU work is then constant in N (100 sequences through the encoder; rows/N x queries*N
          through the MIPS). --scaling strong keeps ONE 100-question batch and splits the encoder slices instead; a
          10 ms step of 2 x 12 dependent transformer layers is latency-bound there.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the MIPS screen kernel mips_screen_kernel<24,1,0>,
HBM-bound; time = HIP events around every search call on the launch stream) and `cpu_baseline`
(oracle/flat_ip_oracle.c, the FAISS-equivalent CPU path, on a bounded row sample).

roofline.traffic is HBM bytes per search call from a separate `rocprofv3 --pmc FETCH_SIZE` pass (scripts/gpu_pmc_screen.sh;
KB x 1024 x 2, the gfx950 correction of MI355X_MICROARCH.md), which cannot run inside this process: the measured ratio
traffic / algorithmic bytes of each kernel (table below, source files under profiles/) is applied to this run's
algorithmic bytes, or pass --pmc-traffic with a fresh measurement.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CHUNK_ROWS = 250_000

# measured HBM fetch bytes per search call / algorithmic bytes (N_pad * d * 4), from rocprofv3 --pmc FETCH_SIZE passes
PMC_TRAFFIC_RATIO = {
    # main 3.75166e6 KB + refine 28996 KB + sample 25367 KB, x2 -> 7.79e9 B at 5M x 768 (15.36e9 B algorithmic)
    "mips_screen_kernel": (0.5072, "profiles/r01_final_mips5m_pmc_screen_FETCH_SIZE.csv"),
    "mips_stream_kernel": (1.001, "profiles/r01_mips1m_pmc_fetch_size.csv"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=5_000_000, help="corpus rows (global)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=100, help="questions per step")
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--topk", type=int, default=1)
    ap.add_argument("--max-q-len", type=int, default=70)
    ap.add_argument("--max-q-sp-len", type=int, default=350)
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32",
                    help="index storage: f32 = fp32-accurate fp16 (hi, lo) pairs (the headline), bf16 = rows rounded to bf16 (BASELINE configs[4])")
    ap.add_argument("--no-encoder", action="store_true", help="MIPS-only step (query embeddings synthetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only for debugging)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = every rank owns its own batch of --batch questions (global batch = batch*N) against the "
                         "row-sharded index; strong = one batch of --batch questions, encoder slices split over ranks")
    ap.add_argument("--share-gpu", action="store_true", help="debug: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--pmc-traffic", type=float, default=None, help="HBM bytes per launch from a rocprofv3 --pmc pass")
    return ap.parse_args()


def corpus_chunk(seed, c, rows, dim, device):
    g = torch.Generator(device=device).manual_seed(seed * 1_000_003 + c)
    return torch.randn((rows, dim), generator=g, device=device, dtype=torch.float32)


def build_shard(index, lo, hi, dim, device, keep_rows=None):
    """Rows [lo, hi) of the global synthetic matrix (chunk-keyed RNG: any sharding reproduces it)."""
    kept = []
    c0, c1 = lo // CHUNK_ROWS, (hi - 1) // CHUNK_ROWS if hi > lo else -1
    for c in range(c0, c1 + 1):
        base = c * CHUNK_ROWS
        blk = corpus_chunk(0, c, CHUNK_ROWS, dim, device)
        a, b = max(lo, base) - base, min(hi, base + CHUNK_ROWS) - base
        index.add(blk[a:b])
        if keep_rows is not None:
            sel = (keep_rows >= base + a) & (keep_rows < base + b)
            kept.append((sel, blk[(keep_rows[sel] - base)].clone()))
        del blk
    return kept


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    from multihop_dense_retrieval_amd import index as mdr_index
    from multihop_dense_retrieval_amd import mhop

    N, d, B = args.rows, args.dim, args.batch
    skw = {"storage": "bf16"} if args.storage == "bf16" else {}
    t0 = time.time()
    if world > 1:
        sidx = mdr_index.ShardedIndexFlatIP(d, N, local_index=mdr_index.IndexFlatIP(d, device=device, **skw))
        lo, hi = sidx.lo, sidx.hi
        local = sidx.local
    else:
        local = mdr_index.IndexFlatIP(d, device=device, **skw)
        sidx = local
        lo, hi = 0, N
    local.reserve(hi - lo)
    # planted hop-1 answers make the run self-checking at full size: question i's best row is p_i
    weak = world > 1 and args.scaling == "weak"
    GB = B * world if weak else B  # questions per step over all ranks
    planted = (torch.arange(GB, device=device, dtype=torch.int64) * 48_611 + 17) % N
    kept = build_shard(local, lo, hi, d, device, keep_rows=planted)
    rows_sum = torch.zeros((GB, d), device=device)
    for sel, rows in kept:
        rows_sum[sel] += rows
    if world > 1:
        dist.all_reduce(rows_sum)
    if weak:  # this rank's own questions
        rows_sum, planted = rows_sum[rank * B:(rank + 1) * B].contiguous(), planted[rank * B:(rank + 1) * B]
    torch.cuda.synchronize()
    build_s = time.time() - t0

    pipe = mhop.SyntheticTwoHop(sidx, batch=B, beam=args.beam, topk=args.topk, dim=d, device=device,
                                max_q_len=args.max_q_len, max_q_sp_len=args.max_q_sp_len,
                                use_encoder=not args.no_encoder, planted_rows=rows_sum, rank=rank, world=world, weak=weak)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.step()
    barrier()
    pipe.reset_kernel_timers()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.step()
    barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # self-check at full size: MIPS-only mode plants the hop-1 answers
    ok = pipe.self_check(out, planted)

    ms_per_step = elapsed / args.steps * 1e3
    qps = GB * args.steps / elapsed
    search_ms = pipe.search_kernel_ms()  # HIP-event average over every timed search call (rank-local)
    stream_bytes = local.stream_bytes()
    # one search call streams the shard once per group of <= 128 queries (the kernel's design point: 8 waves x 16 queries);
    # under weak scaling a rank searches all 100 N queries in its shard, i.e. ceil(100 N / 128) passes per call
    passes = max(1, -(-GB // 128))
    stream_bytes *= passes
    achieved = stream_bytes / (search_ms * 1e-3) / 1e9 if search_ms > 0 else 0.0
    traffic, traffic_src = args.pmc_traffic, "--pmc-traffic"
    if traffic is None:
        for name, (ratio, src) in PMC_TRAFFIC_RATIO.items():
            if name in local.last_kernel() and d == 768 and args.storage != "bf16":
                traffic, traffic_src = float(round(stream_bytes * ratio)), f"{src} (measured ratio {ratio} x algorithmic bytes)"
    roofline = {"bound": "hbm", "kernel": local.last_kernel(), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                "algorithmic_bytes_per_launch": stream_bytes,
                "designed_hbm_bytes_per_launch": stream_bytes // 2 if ("screen" in local.last_kernel() and args.storage != "bf16") else stream_bytes, "avg_launch_ms": round(search_ms, 4),
                "launches_timed": pipe.search_calls_timed(), "corpus_passes_per_launch": passes}

    result = {
        "metric": "queries/sec (2-hop, beam-size x topk) over 5Mx768 index",
        "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
        "dtype": ("bf16 index rows, fp32 accumulate" if args.storage == "bf16" else "f32 (index stored as fp16 hi/lo pairs, fp32 accumulate)")
                 + "; encoder f16 MFMA / f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"synthetic {N}x{d} {'bf16' if args.storage == 'bf16' else 'fp32'} corpus{' row-sharded over ' + str(world) + ' GPUs' if world > 1 else ''}, "
                               f"{B}-question batches{' per GPU (global batch ' + str(GB) + ')' if weak else ''}, 2-hop beam={args.beam} topk={args.topk}"
                               f"{' , MIPS-only (no encoder)' if not pipe.use_encoder else ', RoBERTa-base encoder (random init)'}",
                   "rows": N, "dim": d, "batch": B, "global_batch": GB, "beam": args.beam, "topk": args.topk, "shards": world,
                   "encoder": pipe.encoder_desc(), "index_build_s": round(build_s, 2)},
        "roofline": roofline,
        "self_check": ok,
        "stage_ms": pipe.stage_ms(),
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, device)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, device):
    """FAISS-equivalent CPU path (kind 'port': faiss is not installed here) on the GPU box's host cores, bounded to
    ~args.cpu_seconds: the two searches of one 100-question step over a row SAMPLE of the same synthetic corpus, linearly
    extrapolated to the full row count (flat search is linear in rows). Threads = the cores this container may use
    (affinity and cgroup quota: the boxes give 16 of the host's 256 hardware threads; 128 OpenMP threads on that quota ran
    9x slower). Two restatements of the same algorithm are timed on the sample and the faster one is reported:
    oracle/flat_ip_blas.py (OpenBLAS sgemm behind numpy + top-k, what FAISS itself does) and oracle/flat_ip_oracle.c."""
    import ctypes
    import subprocess

    from oracle import flat_ip_blas
    cores = flat_ip_blas.usable_cpus()
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    f = lib.mdr_oracle_flat_ip_search
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_int]
    d, B, k = args.dim, args.batch, args.beam
    q = corpus_chunk(1, 0, B, d, device).cpu().numpy()
    D = np.empty((B, k), np.float32)
    I = np.empty((B, k), np.int64)

    def run_c(xb):
        t = time.perf_counter()
        rc = f(q.ctypes.data, B, xb.ctypes.data, xb.shape[0], d, k, D.ctypes.data, I.ctypes.data, cores)
        assert rc == 0
        return time.perf_counter() - t, I.copy()

    def run_blas(xb):
        t = time.perf_counter()
        _, Ib = flat_ip_blas.search(q, xb, k)
        return time.perf_counter() - t, Ib

    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cores)
    except Exception:
        limiter = None
    probe = corpus_chunk(0, 0, 100_000, d, device).cpu().numpy()
    run_c(probe), run_blas(probe)  # warm both (thread pools, page faults)
    (tc, Ic), (tb, Ib) = run_c(probe), run_blas(probe)
    agree = float((Ic == Ib).mean())
    rate = 100_000 / min(tb, tc)  # rows/s for one search call
    sample_rows = int(min(args.rows, max(100_000, rate * args.cpu_seconds / 2)))
    sample_rows = min(sample_rows, 4_000_000)  # host RAM bound: 12 GB
    nchunk = -(-sample_rows // CHUNK_ROWS)
    xb = np.concatenate([corpus_chunk(0, c, CHUNK_ROWS, d, device).cpu().numpy() for c in range(nchunk)])[:sample_rows]
    # both restatements on the full sample (the 100k-row probe is too short to rank them reliably on a shared host)
    t_full = {"blas": min(run_blas(xb)[0], run_blas(xb)[0]), "c_openmp": min(run_c(xb)[0], run_c(xb)[0])}
    if t_full["blas"] <= t_full["c_openmp"]:
        run, impl = run_blas, "oracle/flat_ip_blas.py (numpy/OpenBLAS sgemm + top-k, the FAISS algorithm)"
    else:
        run, impl = run_c, "oracle/flat_ip_oracle.c (restatement of faiss IndexFlatIP::search, OpenMP)"
    t1 = min(t_full.values())
    reps = int(max(1, min(10, args.cpu_seconds / max(2 * t1, 1e-3))))  # repeat the (hop 1 + hop 2) pair for a steadier number
    t = sum(run(xb)[0] + run(xb)[0] for _ in range(reps)) / reps
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    step_s = t * (args.rows / sample_rows)
    return {"value": round(B / step_s, 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"2 flat-IP searches (hop 1 + hop 2) of {B} queries, k={k}, over the first {sample_rows} rows of the same "
                      f"synthetic corpus in {t:.2f} s (mean of {reps} repetitions) on {cores} threads (the container's CPU quota; host has {os.cpu_count()} hardware "
                      f"threads), extrapolated linearly to {args.rows} rows; MIPS only (the reference's encoder runs on the GPU in the "
                      f"reference too)",
            "impl": impl, "one_search_over_sample_s": {k_: round(v, 4) for k_, v in t_full.items()}, "probe_id_agreement": agree}


if __name__ == "__main__":
    main()

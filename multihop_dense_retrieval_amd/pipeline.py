"""The batch loop of /root/reference/scripts/eval/eval_mhop_retrieval.py:142-263 as a software pipeline over batches and ranks.

The reference walks one batch at a time on one GPU: tokenise -> encode -> search -> (host) build pairs -> tokenise -> encode ->
search -> rank paths -> metrics, every stage waiting for the previous one. The arithmetic of a batch takes ~6 ms on an MI355X;
tokenising its 100 questions alone takes longer than that on one host core. This module keeps the GPU busy without changing
what any batch computes:

  * host work leaves the GPU's critical path: question tokenisation (and, on the host-tokenizer path, the hop-2 pair
    tokenisation) runs in `--num-workers` forked worker processes (the reference's flag, unused there); path ranking, metrics
    and the JSONL records of batch i-1 are produced by a finisher thread while the GPU runs batch i; every result crosses to
    the host exactly once per batch, through pinned memory, behind an event (no `.cpu()` between the hops on the
    `--hop2-on-device` path);
  * batches are issued in a FIXED order -- hop 1 of batches 0..D-1, then for i = 0, 1, ...: hop 2 of batch i, hop 1 of batch
    i+D -- so the host has D-1 GPU stages of slack for a batch's hop-2 tokenisation, and so that under torch.distributed every
    rank issues its collectives in the same order whatever its host timing is;
  * with W ranks the QUESTIONS are partitioned, not only the index: rank r owns batches r, r+W, r+2W, ...; per hop the ranks'
    query embeddings are all-gathered, every rank searches all W*B queries in its row shard, the per-shard lists are
    all-gathered and merged (ShardedIndexFlatIP.search_device), and each rank continues with the rows of its own questions.
    Per GPU that is 1/W of the encoder forwards and 1/W of the corpus rows; rank 0 collects the records in input order;
  * `fuse` (--pipeline-batches): hop 2 of batch i and hop 1 of batch i+D run as two concurrent encoder forwards (two lanes /
    streams) and share ONE corpus pass.

Every question still walks hop-1 encode -> search -> hop-2 inputs -> encode -> search -> path ranking with the same kernels'
arithmetic; the JSONL of a W-rank run is byte-identical to the one-rank run (tests/test_cli_multirank_gpu.py).
"""
import multiprocessing
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import mhop
from .data import is_roberta_family, roberta_single_np, tokenize_2_11
from .index import all_gather_dim0

# ------------------------------------------------------------------------------------------------------
# tokenizer workers
# ------------------------------------------------------------------------------------------------------
_WORKER_TOK = None


def _worker_init(tokenizer):
    global _WORKER_TOK
    _WORKER_TOK = tokenizer
    os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")  # one batch per worker at a time: parallelism is across workers
    try:
        torch.set_num_threads(1)
    except Exception:
        pass


def _encode_np(tokenizer, texts, pairs, max_lengths):
    """{max_length: {"input_ids": int64 [n, L], "attention_mask": ...[, "token_type_ids"]}} as numpy (crosses the process boundary)."""
    if not (texts if pairs is None else pairs):  # a rank's empty batch (fewer batches than ranks in the last round)
        return {L: {"input_ids": np.zeros((0, L), np.int64), "attention_mask": np.zeros((0, L), np.int64)} for L in max_lengths}
    if is_roberta_family(tokenizer) and pairs is None:  # one BPE pass serves every length (hop-1 cap and the uncapped question of the device path)
        return {L: {"input_ids": i, "attention_mask": m} for L, (i, m) in roberta_single_np(tokenizer, list(texts), max_lengths).items()}
    out = {}
    for L in max_lengths:
        enc = tokenize_2_11(tokenizer, texts, pairs, L)
        out[L] = {k: np.ascontiguousarray(enc[k].numpy() if torch.is_tensor(enc[k]) else np.asarray(enc[k]), dtype=np.int64)
                  for k in ("input_ids", "attention_mask", "token_type_ids") if k in enc}
    return out


def _worker_encode(texts, pairs, max_lengths):
    return _encode_np(_WORKER_TOK, texts, pairs, max_lengths)


class _Now:
    """Future look-alike of an inline call (num_workers = 0)."""

    def __init__(self, fn, *a):
        self._v = fn(*a)

    def get(self, timeout=None):
        return self._v


_FINISH_FN = None


def _finish_init(fn):
    global _FINISH_FN
    _FINISH_FN = fn
    try:
        torch.set_num_threads(1)
    except Exception:
        pass


def _finish_task(ann, D, I, D2, I2):
    return _FINISH_FN(ann, D, I, D2, I2)


def fork_workers_allowed(workers, what):
    """Worker processes are forked, and a fork must come BEFORE the process touches the HIP device: a child that inherits a live runtime (and the
    Pool's handler threads of an earlier call) is the classic fork-with-threads hazard, and every later device allocation of the parent got ~50x slower
    (measured). The CLI keeps that order on its first call; a SECOND in-process call (tests, notebooks) arrives with the device
    already initialised: it runs with workers = 0 (same results, inline) and says so, unless MDR_ALLOW_LATE_FORK=1."""
    workers = max(0, int(workers))
    if workers > 0 and torch.cuda.is_available() and torch.cuda.is_initialized() and os.environ.get("MDR_ALLOW_LATE_FORK", "0") != "1":
        import warnings
        warnings.warn(f"{what}: the HIP runtime is already initialised in this process; not forking {workers} worker process(es) after it -- running "
                      f"inline (workers = 0). Start a fresh process, or set MDR_ALLOW_LATE_FORK=1.", RuntimeWarning, stacklevel=3)
        return 0
    return workers


class FinishPool:
    """Path ranking / metrics / output records of a batch are pure Python over id2doc: in the issuing process they fight the launching thread for
    the GIL (measured: 13 ms of finisher time per batch and a 2x slower loop). `workers` forked processes run `finish` instead; they are forked AFTER
    the corpus is loaded (they inherit id2doc and the finish closure) and never touch the device. workers = 0: finish runs on the finisher thread."""

    def __init__(self, finish, workers):
        self.finish, self.pool = finish, None
        workers = fork_workers_allowed(workers, "FinishPool")
        if workers > 0:
            self.pool = multiprocessing.get_context("fork").Pool(int(workers), initializer=_finish_init, initargs=(finish,))

    def submit(self, ann, D, I, D2, I2):
        """Arrays are copied (the pinned staging buffers they view are recycled). -> handle with .get()."""
        args = (ann, np.array(D), np.array(I), np.array(D2), np.array(I2))
        if self.pool is None:
            return _Now(self.finish, *args)
        return self.pool.apply_async(_finish_task, args)

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None


class TokenizerPool:
    """`workers` forked processes that hold the tokenizer; `submit` returns a handle with .get(). MUST be created before the
    process touches the HIP device (a forked child must not inherit a live runtime); workers = 0 tokenises inline."""

    def __init__(self, tokenizer, workers):
        self.tokenizer = tokenizer
        self.workers = fork_workers_allowed(workers, "TokenizerPool")
        self.pool = None
        self._inline_lock = threading.Lock()  # HF fast tokenizers are not re-entrant ("Already borrowed"): inline calls come from several threads
        if self.workers > 0:
            self.pool = multiprocessing.get_context("fork").Pool(self.workers, initializer=_worker_init, initargs=(tokenizer,))

    def submit(self, texts, pairs, max_lengths):
        if self.pool is None:
            with self._inline_lock:
                return _Now(_encode_np, self.tokenizer, texts, pairs, tuple(max_lengths))
        return self.pool.apply_async(_worker_encode, (texts, pairs, tuple(max_lengths)))

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None


# ------------------------------------------------------------------------------------------------------
# the pipeline
# ------------------------------------------------------------------------------------------------------
class _Job:
    __slots__ = ("idx", "lo", "hi", "n", "questions", "ann", "tok1", "q_ids", "q_mask", "D", "I", "D_host", "I_host", "e1", "host", "e2", "result")

    def __init__(self, idx, lo, hi, questions, ann):
        self.idx, self.lo, self.hi, self.n = idx, lo, hi, hi - lo
        self.questions, self.ann = questions, ann
        self.tok1 = self.q_ids = self.q_mask = self.D = self.I = self.D_host = self.I_host = self.e1 = self.host = self.e2 = self.result = None


class TwoHopPipeline:
    """run(questions, ds_items) -> list of per-batch results (this rank's batches; `finish(ann, D, I, D2, I2)` makes one).

    model.encode_q(ids, mask, type_ids[, lane]) -> [n, d]; index.search_device(q, k) -> (D, I) device tensors (IndexFlatIP, or
    ShardedIndexFlatIP whose search_device runs the exchange); arena = TokenArena for device-side hop-2 assembly or None for the
    reference's host path (id2doc lookups + pair tokenisation)."""

    PAIR_CHUNK = 25  # (question, passage) pairs per tokenizer task

    def __init__(self, model, index, pool, id2doc, finish, *, batch_size, beam, max_q_len, max_q_sp_len, roberta=True, arena=None,
                 device=None, rank=0, world=1, group=None, depth=None, fuse=False, finish_workers=0, finish_pool=None):
        self.model, self.index, self.pool, self.id2doc, self.finish = model, index, pool, id2doc, finish
        # finish_pool: made by the caller BEFORE it touched the device (the CLI); finish_workers: made here (tests, tools)
        self._own_finish_pool = finish_pool is None
        self.finish_pool = finish_pool if finish_pool is not None else FinishPool(finish, finish_workers)
        self.B, self.beam, self.Lq, self.Lsp, self.roberta = int(batch_size), int(beam), int(max_q_len), int(max_q_sp_len), roberta
        self.arena = arena
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.cuda = self.device.type == "cuda"
        self.rank, self.world, self.group = int(rank), int(world), group
        # in-flight depth: the host path needs slack for the pair tokenisation between the hops; the device path only prefetches
        self.depth = int(depth) if depth else (2 if arena is not None else 8)
        self.fuse = bool(fuse)
        self.d = None
        self._side = None
        self._lanes = self._has_lanes(model)
        self.stats = {"batches": 0, "hop1_forwards": 0, "hop2_forwards": 0, "searches": 0, "queries_searched": 0, "batch_done_t": [],
                      "wait_tok1_s": 0.0, "wait_tok2_s": 0.0, "finish_busy_s": 0.0, "drain_s": 0.0, "main_s": {}, "gpu_stage_ms": None}
        self._stage_ev = []
        self._h2d_ring, self._d2h_free, self._d2h_lock = {}, {}, threading.Lock()

    def close(self):
        if self._own_finish_pool:
            self.finish_pool.close()

    @staticmethod
    def _has_lanes(model):
        import inspect
        try:
            return "lane" in inspect.signature(model.encode_q).parameters
        except (TypeError, ValueError):
            return False

    # -- host <-> device plumbing ---------------------------------------------------------------------------
    # Pinned staging buffers are allocated ONCE per (shape, dtype) and recycled: a fresh `torch.empty(pin_memory=True)` per transfer cost
    # 2-3 ms each on the MI355X boxes (hipHostMalloc; measured: 10 ms of a 13.7 ms batch went there), more than a batch's tokenisation.
    def _h2d(self, arr):
        t = torch.from_numpy(arr)
        if not self.cuda:
            return t
        tt = time.perf_counter()
        key = (tuple(t.shape), t.dtype)
        ring = self._h2d_ring.setdefault(key, [])
        slot = None
        for cand in ring:  # a buffer whose previous copy has left the host
            if cand[1].query():
                slot = cand
                break
        if slot is None:
            if len(ring) >= 16:
                slot = ring[0]
                slot[1].synchronize()
            else:
                slot = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True), None]
                ring.append(slot)
        # (numpy's single-threaded memcpy: torch's CPU copy_ forks its intra-op pool -- 256 threads on the MI355X hosts, 3-4 ms per 280 KB copy measured)
        np.copyto(slot[0].numpy(), arr)
        dev = slot[0].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        self._tick("h2d", tt)
        for i, cand in enumerate(ring):  # least recently used first (identity, not ==: the slots hold tensors)
            if cand is slot:
                ring.append(ring.pop(i))
                break
        return dev

    def _d2h(self, t):
        """Device tensor -> pinned host tensor (async; valid behind the stage's event). The buffer returns to the pool in _finish."""
        if not self.cuda:
            return t.clone()
        key = (tuple(t.shape), t.dtype)
        with self._d2h_lock:
            free = self._d2h_free.setdefault(key, [])
            host = free.pop() if free else None
        if host is None:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        return host

    def _release(self, *hosts):
        if not self.cuda:
            return
        with self._d2h_lock:
            for h in hosts:
                if h is not None:
                    self._d2h_free.setdefault((tuple(h.shape), h.dtype), []).append(h)

    def _event(self):
        if not self.cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _encode(self, enc, lane=0):
        ids, mask, tt = enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids")
        if ids.shape[0] == 0:
            return torch.zeros((0, self._dim()), dtype=torch.float32, device=self.device)
        if self._lanes:
            return self.model.encode_q(ids, mask, tt, lane=lane)
        return self.model.encode_q(ids, mask, tt)

    def _dim(self):
        if self.d is None:
            self.d = int(getattr(self.index, "d"))
        return self.d

    # -- search: one rank, or all ranks' queries against every shard ---------------------------------------------
    def _search(self, q, n_pad):
        """q [n, d]: this rank's queries (n <= n_pad). -> (D, I) [n, beam] of those queries over the WHOLE corpus."""
        self.stats["searches"] += 1
        if self.world == 1:
            self.stats["queries_searched"] += int(q.shape[0])
            if q.shape[0] == 0:
                return (torch.zeros((0, self.beam), dtype=torch.float32, device=self.device), torch.zeros((0, self.beam), dtype=torch.int64, device=self.device))
            return self.index.search_device(q.contiguous(), self.beam)
        n = int(q.shape[0])
        qp = q
        if n < n_pad:  # a fixed block per rank: the collective's shape must not depend on a ragged last batch
            qp = torch.zeros((n_pad, q.shape[1]), dtype=q.dtype, device=q.device)
            qp[:n] = q
            # pad rows are searched by every rank and thrown away: an all-zero query would tie EVERY corpus row at score 0, overflow the screen kernels'
            # candidate lists and send the whole call (the real queries too) through the exact fallback pass. A pad row is a copy of this
            # rank's last real query, or the first basis vector on a rank without one (scores = column 0 of the corpus: generic, tie-free).
            if n > 0:
                qp[n:] = q[n - 1]
            else:
                qp[:, 0] = 1.0
        allq = all_gather_dim0(qp.contiguous(), self.world, self.group)
        self.stats["queries_searched"] += int(allq.shape[0])
        D, I = self.index.search_device(allq, self.beam)
        lo = self.rank * n_pad
        return D[lo:lo + n].contiguous(), I[lo:lo + n].contiguous()

    # -- stages ---------------------------------------------------------------------------------------------------
    def _tok1(self, job):
        lens = (self.Lq, self.Lsp) if self.arena is not None and self.Lsp != self.Lq else (self.Lq,)
        job.tok1 = self.pool.submit(job.questions, None, lens)

    def _hop1_inputs(self, job):
        t0 = time.perf_counter()
        enc = job.tok1.get()
        self.stats["wait_tok1_s"] += time.perf_counter() - t0
        job.tok1 = None
        e1 = {k: self._h2d(v) for k, v in enc[self.Lq].items()}
        if self.arena is not None:
            # the question without the hop-1 length cap, so that the pair sees the tokens the tokenizer would (eval_mhop_retrieval.py:168)
            full = enc[self.Lsp] if self.Lsp != self.Lq else enc[self.Lq]
            job.q_ids, job.q_mask = (e1["input_ids"], e1["attention_mask"]) if self.Lsp == self.Lq else (self._h2d(full["input_ids"]), self._h2d(full["attention_mask"]))
        return e1

    def _after_hop1(self, job, D, I):
        job.D, job.I = D, I
        if self.arena is None:  # host path: the ids go to the host, a thread builds and tokenises the pairs
            job.D_host, job.I_host = self._d2h(D), self._d2h(I)
            job.e1 = self._event()
            job.host = self._threads.submit(self._make_pairs, job)

    def hop1(self, job):
        t = time.perf_counter()
        enc = self._hop1_inputs(job)
        t = self._tick("hop1_inputs", t)
        q = self._encode(enc)
        t = self._tick("encode_issue", t)
        self.stats["hop1_forwards"] += int(job.n > 0)
        D, I = self._search(q, self.B)
        t = self._tick("search_issue", t)
        self._after_hop1(job, D, I)
        self._tick("after", t)

    def _make_pairs(self, job):
        """Worker thread: wait for the hop-1 lists, look the passages up (eval_mhop_retrieval.py:158-166), tokenise the pairs."""
        if job.e1 is not None:
            job.e1.synchronize()
        D, I = job.D_host.numpy(), job.I_host.numpy()
        if job.n == 0:
            return None
        pairs = mhop.build_hop2_pairs(job.questions, D, I, self.id2doc, roberta=self.roberta)  # D gets the -inf of empty passages
        # a batch's pairs go to the workers in pieces: its tokenisation latency (what the in-flight depth has to cover) shrinks with the piece
        step = self.PAIR_CHUNK if getattr(self.pool, "workers", 0) > 1 else len(pairs)
        parts = [self.pool.submit(None, pairs[lo:lo + step], (self.Lsp,)) for lo in range(0, len(pairs), step)]
        parts = [p.get()[self.Lsp] for p in parts]
        return {k: np.concatenate([p[k] for p in parts], 0) for k in parts[0]}

    def _hop2_inputs(self, job):
        if self.arena is not None:
            if job.n == 0:
                z = torch.zeros((0, self.Lsp), dtype=torch.int64, device=self.device)
                return {"input_ids": z, "attention_mask": z}
            ids2, mask2 = self.arena.assemble_hop2(job.q_ids, job.q_mask, job.I, job.D, self.Lsp)  # D: -inf of empty passages, in place
            job.q_ids = job.q_mask = None
            return {"input_ids": ids2, "attention_mask": mask2}
        t0 = time.perf_counter()
        enc = job.host.result()
        self.stats["wait_tok2_s"] += time.perf_counter() - t0
        job.host = None
        if enc is None:
            z = torch.zeros((0, self.Lsp), dtype=torch.int64, device=self.device)
            return {"input_ids": z, "attention_mask": z}
        return {k: self._h2d(v) for k, v in enc.items()}

    def _after_hop2(self, job, D2, I2):
        if self.arena is not None:
            job.D_host, job.I_host = self._d2h(job.D), self._d2h(job.I)
        d2, i2 = self._d2h(D2), self._d2h(I2)
        job.D = job.I = None
        job.e2 = self._event()
        job.result = self._finisher.submit(self._finish, job, d2, i2)

    def hop2(self, job):
        t = time.perf_counter()
        ev0 = self._event()
        enc = self._hop2_inputs(job)
        t = self._tick("hop2_inputs", t)
        q2 = self._encode(enc)
        t = self._tick("encode_issue", t)
        self.stats["hop2_forwards"] += int(job.n > 0)
        D2, I2 = self._search(q2, self.B * self.beam)
        t = self._tick("search_issue", t)
        self._after_hop2(job, D2, I2)
        self._tick("after", t)
        if ev0 is not None:
            self._stage_ev.append((ev0, job.e2))

    def _tick(self, name, t0):
        """Main-thread seconds per section of a stage (stats["main_s"]): where the issuing thread's time goes."""
        t1 = time.perf_counter()
        self.stats["main_s"][name] = self.stats["main_s"].get(name, 0.0) + (t1 - t0)
        return t1

    def hop2_and_hop1(self, job, nxt):
        """hop 2 of `job` beside hop 1 of `nxt`: two concurrent forwards (two lanes on two streams), ONE corpus pass for both."""
        t = time.perf_counter()
        ev0 = self._event()
        enc2 = self._hop2_inputs(job)
        t = self._tick("hop2_inputs", t)
        enc1 = self._hop1_inputs(nxt)
        t = self._tick("hop1_inputs", t)
        if self.cuda and self._lanes:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            start = torch.cuda.Event()
            start.record()
            self._side.wait_event(start)
            with torch.cuda.stream(self._side):
                q1 = self._encode(enc1, lane=1)
                done = torch.cuda.Event()
                done.record()
            q2 = self._encode(enc2)
            torch.cuda.current_stream().wait_event(done)
            q1.record_stream(torch.cuda.current_stream())
        else:
            q1 = self._encode(enc1)
            q2 = self._encode(enc2)
        t = self._tick("encode_issue", t)
        self.stats["hop1_forwards"] += int(nxt.n > 0)
        self.stats["hop2_forwards"] += int(job.n > 0)
        n2, nb2 = int(q2.shape[0]), self.B * self.beam
        if self.world == 1:
            Dc, Ic = self._search(torch.cat([q2, q1], 0), nb2 + self.B)
            D2, I2, D1, I1 = Dc[:n2], Ic[:n2], Dc[n2:], Ic[n2:]
        else:  # fixed layout inside this rank's block: hop-2 rows at 0, hop-1 rows at B*beam
            blk = torch.zeros((nb2 + self.B, q2.shape[1] if n2 else q1.shape[1]), dtype=torch.float32, device=self.device)
            blk[:n2] = q2
            blk[nb2:nb2 + q1.shape[0]] = q1
            Dc, Ic = self._search(blk, nb2 + self.B)
            D2, I2, D1, I1 = Dc[:n2], Ic[:n2], Dc[nb2:nb2 + q1.shape[0]], Ic[nb2:nb2 + q1.shape[0]]
        t = self._tick("search_issue", t)
        self._after_hop2(job, D2.contiguous(), I2.contiguous())
        self._after_hop1(nxt, D1.contiguous(), I1.contiguous())
        t = self._tick("after", t)
        if ev0 is not None:
            self._stage_ev.append((ev0, job.e2))

    def _finish(self, job, d2, i2):
        if job.e2 is not None:
            job.e2.synchronize()
        r = None
        if job.n > 0:
            t0 = time.perf_counter()
            r = self.finish_pool.submit(job.ann, job.D_host.numpy(), job.I_host.numpy(), d2.numpy(), i2.numpy())
            self.stats["finish_busy_s"] += time.perf_counter() - t0
            self.stats["batch_done_t"].append(time.perf_counter())  # the batch's results are on the host
        self._release(job.D_host, job.I_host, d2, i2)
        job.D_host = job.I_host = None
        return r

    # -- the loop ---------------------------------------------------------------------------------------------------
    def run(self, questions, ds_items):
        nb = -(-len(questions) // self.B) if questions else 0
        rounds = -(-nb // self.world) if nb else 0
        jobs = []
        for g in range(rounds):
            b = g * self.world + self.rank
            lo, hi = min(len(questions), b * self.B), min(len(questions), (b + 1) * self.B)
            jobs.append(_Job(b, lo, hi, questions[lo:hi], ds_items[lo:hi]))
        D = max(1, self.depth)
        ahead = D + max(2, min(16, 2 * getattr(self.pool, "workers", 0)))  # tokenised this many batches before their hop 1 is issued
        self._threads = ThreadPoolExecutor(max_workers=max(2, D + 1), thread_name_prefix="mdr-host")
        self._finisher = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mdr-finish")  # one thread: results complete in batch order
        # the issuing thread shares the GIL with the finisher and the pair builders: with CPython's default 5 ms switch interval every launch
        # it makes while one of them computes waits up to 5 ms for the lock -- most of a batch's GPU time
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)
        try:
            for j in jobs[:ahead]:
                self._tok1(j)
            with torch.no_grad():
                for g in range(min(D, rounds)):
                    self.hop1(jobs[g])
                    if g + ahead < rounds:
                        self._tok1(jobs[g + ahead])
                for g in range(rounds):
                    nxt = jobs[g + D] if g + D < rounds else None
                    if nxt is not None and self.fuse:
                        self.hop2_and_hop1(jobs[g], nxt)
                    else:
                        self.hop2(jobs[g])
                        if nxt is not None:
                            self.hop1(nxt)
                    if g + D + ahead < rounds:
                        self._tok1(jobs[g + D + ahead])
                    self.stats["batches"] += int(jobs[g].n > 0)
            out = []
            t0 = time.perf_counter()
            for j in jobs:
                r = j.result.result()
                j.result = None
                if r is not None:
                    out.append((j.idx, r.get()))
            self.stats["drain_s"] = time.perf_counter() - t0  # issuing done -> last batch finished on the host
            if self._stage_ev:  # device time of the hop-2 stages (issue of the stage's first op -> its last D2H done), steady state: median
                ms = sorted(a.elapsed_time(b) for a, b in self._stage_ev)
                self.stats["gpu_stage_ms"] = {"median": round(ms[len(ms) // 2], 3), "min": round(ms[0], 3), "max": round(ms[-1], 3)}
            return out
        finally:
            sys.setswitchinterval(old_switch)
            self._threads.shutdown(wait=True)
            self._finisher.shutdown(wait=True)


def gather_results(per_batch, world, group=None):
    """[(batch index, result)] of every rank -> on rank 0 the results of ALL batches in input order (others: None)."""
    if world == 1:
        return [r for _, r in sorted(per_batch, key=lambda t: t[0])]
    import torch.distributed as dist
    rank = dist.get_rank(group)
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(per_batch, bucket, dst=0, group=group)
    if rank != 0:
        return None
    merged = sorted((t for part in bucket for t in part), key=lambda t: t[0])
    return [r for _, r in merged]

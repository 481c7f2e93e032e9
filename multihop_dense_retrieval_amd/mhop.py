"""Two-hop beam-search retrieval: the host-side logic of
/root/reference/scripts/eval/eval_mhop_retrieval.py:139-284 as importable functions (the reference
keeps it inline under `__main__`), plus the device-resident variant used by bench.py.

Everything numeric (encoder forward, MIPS) is delegated to libmdrhip.so through `retriever.py` and
`index.py`; this module only wires the hops together, ranks paths and computes the metrics.
"""
import collections
import json

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------
# host logic shared with the CLI (parity-tested against oracle/mhop_oracle.py and tests/golden/mhop.json)
# ------------------------------------------------------------------------------------------------------
def strip_question(q):
    """Drop exactly one trailing '?' before encoding (eval_mhop_retrieval.py:139)."""
    return q[:-1] if q.endswith("?") else q


def load_corpus_dict(path_or_obj):
    """id2doc with string keys -> {"title","text",...}; list-valued entries [title, text, (intro)] are
    normalised (eval_mhop_retrieval.py:131-133)."""
    id2doc = path_or_obj
    if isinstance(path_or_obj, str):
        if path_or_obj.endswith(".store"):  # memory-mapped offset-table form of the same data (corpus_store.py)
            from .corpus_store import CorpusStore
            return CorpusStore(path_or_obj)
        with open(path_or_obj) as f:
            id2doc = json.load(f)
    if len(id2doc) and isinstance(next(iter(id2doc.values())), list):
        id2doc = {k: {"title": v[0], "text": v[1]} for k, v in id2doc.items()}
    return id2doc


def build_hop2_pairs(batch_q, D, I, id2doc, roberta=True):
    """(question, passage) pairs for the hop-2 encoder, row-major over (question, beam slot). A passage
    with empty text is replaced by its title and its hop-1 score set to -inf, in place
    (eval_mhop_retrieval.py:158-166)."""
    pairs = []
    ninf = float("-inf")
    for b, q in enumerate(batch_q):
        row = I[b]
        for j in range(len(row)):
            doc = id2doc[str(int(row[j]))]
            text = doc["text"]
            if roberta and not text.strip():
                text = doc["title"]
                D[b][j] = ninf
            pairs.append((q, text))
    return pairs


def rank_paths(D, I, D2, I2, beam, topk, beam2=None):
    """Best `topk` (hop-1 id, hop-2 id, score) chains per question (eval_mhop_retrieval.py:181-206; with separate hop
    widths `beam` x `beam2`, eval_mhop_fever.py:111-130). Path score = hop-1 score + hop-2 score. Raises IndexError when
    topk > beam*beam2, like the reference. Among equal path scores the reference's order is unspecified (reversed
    unstable argsort); here it is the same numpy expression so results coincide on every input."""
    B = D.shape[0]
    beam2 = beam if beam2 is None else beam2
    if topk > beam * beam2:
        raise IndexError(f"topk={topk} exceeds beam*beam2={beam * beam2}")
    scores = (np.asarray(D)[:, :, None] + np.asarray(D2).reshape(B, beam, beam2)).reshape(B, beam * beam2)
    I2 = np.asarray(I2).reshape(B, beam, beam2)
    order = np.argsort(scores, axis=1)[:, ::-1][:, :topk]
    out = []
    for b in range(B):
        i, j = np.divmod(order[b], beam2)
        out.append([(int(I[b, ii]), int(I2[b, ii, jj]), float(scores[b, o])) for ii, jj, o in zip(i, j, order[b])])
    return out


def question_metrics(chains, sp, id2doc):
    """p_recall / p_em / recall_1 / path_covered on titles (eval_mhop_retrieval.py:219-242)."""
    if len(set(sp)) != 2:
        raise AssertionError("expected exactly two distinct supporting titles")
    path_titles = [(id2doc[str(h1)]["title"], id2doc[str(h2)]["title"]) for h1, h2, _ in chains]
    retrieved = {t for p in path_titles for t in p}
    hop1 = {p[0] for p in path_titles}
    hit = [t in retrieved for t in sp]
    gold = set(sp)
    return {"p_recall": int(any(hit)), "p_em": int(all(hit)), "recall_1": int(any(t in hop1 for t in sp)),
            "path_covered": int(any(set(p) == gold for p in path_titles))}


def output_record(item, chains, id2doc):
    """One JSONL record (eval_mhop_retrieval.py:246-258): keys in this order, original question text."""
    return {"_id": item["_id"], "question": item["question"],
            "candidate_chains": [[id2doc[str(h1)], id2doc[str(h2)]] for h1, h2, _ in chains]}


def summary_lines(metrics):
    """The log lines of eval_mhop_retrieval.py:265-284 (retrieval branch), verbatim format."""
    groups = collections.OrderedDict()
    for m in metrics:
        groups.setdefault(m["type"], []).append(m)

    def block(ms):
        return [f"\tAvg PR: {np.mean([m['p_recall'] for m in ms])}", f"\tAvg P-EM: {np.mean([m['p_em'] for m in ms])}",
                f"\tAvg 1-Recall: {np.mean([m['recall_1'] for m in ms])}", f"\tPath Recall: {np.mean([m['path_covered'] for m in ms])}"]

    lines = [f"Evaluating {len(metrics)} samples..."] + block(metrics)
    for t, ms in groups.items():
        lines.append(f"{t} Questions num: {len(ms)}")
        lines += block(ms)
    return lines


def rank_paths_device(D, I, D2, I2, beam, topk):
    """Device version of rank_paths for the resident pipeline: tensors in, tensors out, no sync.
    D [B,beam], I [B,beam], D2/I2 [B*beam, beam] -> (hop1 [B,topk], hop2 [B,topk], score [B,topk])."""
    B = D.shape[0]
    scores = (D[:, :, None] + D2.view(B, beam, beam)).view(B, beam * beam)
    s, o = torch.topk(scores, topk, dim=1)
    i = torch.div(o, beam, rounding_mode="floor")
    h1 = torch.gather(I, 1, i)
    h2 = torch.gather(I2.view(B, beam * beam), 1, o)
    return h1, h2, s


# ------------------------------------------------------------------------------------------------------
# bench pipeline: everything resident in HBM, synthetic inputs (no tokenizer / corpus text offline)
# ------------------------------------------------------------------------------------------------------
class SyntheticTwoHop:
    """One `step()` = one batch of questions through hop-1 encode -> search -> hop-2 input assembly ->
    hop-2 encode -> search -> path ranking, without leaving the device. Token ids follow SURVEY.md
    §8(d): uniform in [3, vocab), <s>=0 first, </s>=2 last, pad 1, question lengths U[8,40], passage
    lengths U[60,300]; hop-2 inputs are gathered from a synthetic token arena by the hop-1 doc ids."""

    VOCAB = 50265

    def __init__(self, index, batch, beam, topk, dim, device, max_q_len=70, max_q_sp_len=350, use_encoder=True,
                 planted_rows=None, rank=0, world=1, weak=False, pipelined=False, pool=1):
        """`weak=False`: ONE batch of `batch` questions shared by all ranks (strong scaling: the encoder work is split).
        `weak=True`: every rank owns its own batch of `batch` questions (global batch = batch * world): it encodes them,
        the embeddings of all ranks are all-gathered, every rank searches ALL of them in its row shard, the per-shard
        lists are all-gathered and merged, and the rank continues with the rows of its own questions."""
        self.index, self.B, self.beam, self.topk, self.d, self.device = index, batch, beam, topk, dim, device
        self.Lq, self.Lsp = max_q_len, max_q_sp_len
        self.use_encoder = use_encoder
        self.pipelined = bool(pipelined)
        self._ready = collections.deque()  # pipelined mode: (q, D, I) of the upcoming batches whose hop 1 is already done, in order
        self._side = None   # side stream of the pipelined loop
        self._search_stream = None
        self.rank, self.world = rank, world
        self.weak = bool(weak) and world > 1
        self.local = getattr(index, "local", index)
        g = torch.Generator(device=device).manual_seed(2 + (1000 * rank if self.weak else 0))
        B = batch
        # `pool` DIFFERENT question batches, walked round-robin by step(): different lengths and tokens -> different hop-1 answers ->
        # different hop-2 token totals (the GEMM tile counts move with them); pool = 1 repeats one batch (rounds 1-3)
        self.pool = max(1, int(pool))
        self.batches = []
        pos = torch.arange(self.Lq, device=device)[None, :]
        for _ in range(self.pool):
            q_len = torch.randint(8, 41, (B,), generator=g, device=device)
            q_ids = torch.randint(3, self.VOCAB, (B, self.Lq), generator=g, device=device)
            q_mask = (pos < q_len[:, None]).long()
            q_ids = torch.where(pos == 0, torch.zeros_like(q_ids), q_ids)
            q_ids = torch.where(pos == q_len[:, None] - 1, torch.full_like(q_ids, 2), q_ids)
            q_ids = torch.where(q_mask.bool(), q_ids, torch.ones_like(q_ids))
            noise = 0.05 * torch.randn((B, dim), generator=g, device=device)
            self.batches.append({"q_len": q_len, "q_ids": q_ids, "q_mask": q_mask, "noise": noise})
        self._cur = 0  # pool entry of the batch step() works on (pipelined: the batch whose hop 2 runs)
        self.table = torch.randn((1024, dim), generator=g, device=device)
        self.step_log = []  # per step: (pool entry, hop-2 sequence lengths [B*beam] as a device tensor or None)
        self.planted_rows = planted_rows
        self.encoder = None
        self.arena = None
        if use_encoder and not getattr(SyntheticTwoHop, "_defer_encoder", False):
            from .arena import TokenArena
            from .retriever import RobertaRetriever
            self.encoder = RobertaRetriever.random_init(device=device, seed=3)
            self.arena = TokenArena.synthetic(int(index.ntotal), device, seed=5, vocab=self.VOCAB)
        self._ev = []
        self._search_ev = []

    # -- the current batch of the pool ------------------------------------------------------------------------
    @property
    def q_len(self):
        return self.batches[self._cur]["q_len"]

    @property
    def q_ids(self):
        return self.batches[self._cur]["q_ids"]

    @property
    def q_mask(self):
        return self.batches[self._cur]["q_mask"]

    @property
    def noise(self):
        return self.batches[self._cur]["noise"]

    def _nxt(self):
        return self.batches[(self._cur + 1) % self.pool]

    # -- hop-2 inputs: assembled on the device from the (synthetic) token arena -----------------------------
    def _hop2_inputs(self, I, D=None):
        """`<s> q </s></s> passage </s>` with longest-first truncation to max_q_sp_len (mdr_assemble_hop2)."""
        return self.arena.assemble_hop2(self.q_ids, self.q_mask, I, D, self.Lsp)

    def _own(self, t):
        """Rows of this rank's own questions out of a [world * n, ...] gathered tensor."""
        n = t.shape[0] // self.world
        return t[self.rank * n:(self.rank + 1) * n]

    def _interleave(self, a, b):
        """Weak scaling: both tensors are all-gathered per rank ([world * na], [world * nb]); the fused search wants each
        rank's rows together: [rank][na + nb]."""
        w = self.world
        na, nb = a.shape[0] // w, b.shape[0] // w
        return torch.cat([a.view(w, na, -1), b.view(w, nb, -1)], 1).reshape(w * (na + nb), -1)

    def _encode(self, ids, mask, lane=0):
        if self.weak:
            from .index import all_gather_dim0
            return all_gather_dim0(self.encoder.encode_q(ids, mask, None, lane=lane), self.world)
        if self.world > 1:
            # data-parallel encoder (strong scaling): rank r encodes rows r, r + W, r + 2W, ... and the embeddings are all-gathered.
            # The split is INTERLEAVED, not contiguous, so that the ranks' TOKEN counts (the encoder's cost; hop-2 rows are 64..350
            # tokens long) balance without a host sync on the lengths -- an exact split by cumulative token count would need the
            # lengths on the host between the hops. Static row counts per rank also keep the hipGraph shapes fixed.
            n = ids.shape[0]
            per = -(-n // self.world)
            mine = torch.arange(self.rank, n, self.world, device=ids.device)
            part = torch.zeros((per, self.d), device=self.device)
            if mine.numel() > 0:
                part[: mine.numel()] = self.encoder.encode_q(ids[mine], mask[mine], None, lane=lane)
            from .index import all_gather_dim0
            g = all_gather_dim0(part, self.world)  # [W * per, d]: entry (r, j) is global row j * W + r
            return g.view(self.world, per, self.d).transpose(0, 1).reshape(self.world * per, self.d)[:n].contiguous()
        return self.encoder.encode_q(ids, mask, None, lane=lane)

    # -- one step ----------------------------------------------------------------------------------------
    def _mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _search(self, q, k):
        e0 = self._mark()
        if self.world == 1:
            D, I = self.local.search_device(q, k)
            self._search_ev.append((e0, self._mark(), int(q.shape[0])))
            return D, I
        # N > 1: the shard's lists land in the packed exchange block, ONE all-gather moves every rank's block, ONE kernel merges them
        block, _, _ = self.local.search_device_packed(q, k)
        self._search_ev.append((e0, self._mark(), int(q.shape[0])))  # the local MIPS launch only (roofline), not the exchange
        return self.index.exchange_packed(block, int(q.shape[0]), k)

    # -- software-pipelined step ------------------------------------------------------------------------------
    def _hop1_only(self):
        """Prologue of the pipelined loop: hop 1 of the first batch."""
        if self.use_encoder:
            q = self._encode(self.q_ids, self.q_mask)
        else:
            q = self.planted_rows + self.noise
            if self.weak:
                from .index import all_gather_dim0
                q = all_gather_dim0(q, self.world)
        D, I = self._search(q, self.beam)
        if self.weak:
            q, D, I = self._own(q), self._own(D).contiguous(), self._own(I).contiguous()
        return q, D, I

    def _step_pipelined(self):
        """Batches are independent, so hop 2 of batch i and hop 1 of batch i+1 run as TWO CONCURRENT encoder forwards (two lanes
        = two workspaces and graph caches on two streams, the same weights; one merged 22.6 k-token forward measured slower,
        NEGATIVE_RESULTS.md §6) and share ONE fused corpus pass (B*beam + B queries; more than 128 queries go 256 per pass). Every
        question still walks hop-1 encode -> search -> hop-2 assembly -> hop-2 encode -> search -> path ranking with the same
        arithmetic; what changes is that the small, latency-bound hop-1 forward runs beside the previous batch's large hop-2
        forward and that its queries ride in that batch's corpus pass. In the steady state one call finishes one batch and
        starts the next, i.e. per step exactly one hop-1 and one hop-2 of every kind of work, as in the sequential step.
        `stage_ms()["hop2_encode"]` of this loop is the wall time of BOTH forwards (main stream, waits for the side stream)."""
        if not self._ready:
            self._ready.append(self._hop1_only())
            self._search_ev = self._search_ev[:-1] if self._search_ev else self._search_ev  # the prologue is not a timed call
        q, D, I = self._ready.popleft()
        B, bm = self.B, self.beam
        nxt = [self._nxt()]  # this step carries the hop 1 of the next batch
        ev = [self._mark()]
        ev.append(ev[0])  # (no separate hop-1 stages)
        ev.append(ev[0])
        q_next = None
        if self.use_encoder:
            done = None
            # the next batch's questions on a side stream / second encoder lane, beside this batch's hop-2 forward: short, latency-bound
            # launches that fill the gaps of the large forward instead of running alone (measured alternatives -- the questions of several
            # batches as one forward, the small forward beside the corpus pass, two batches in flight: scripts/measure/loop_variants.py)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)  # (a high-priority side stream measured the same: 7.21 vs 7.26 ms)
            start = torch.cuda.Event()
            start.record()
            self._side.wait_event(start)
            n_ids, n_mask = nxt[0]["q_ids"], nxt[0]["q_mask"]
            with torch.cuda.stream(self._side):
                q_next = self._encode(n_ids, n_mask, lane=1)
                done = torch.cuda.Event()
                done.record()
            n_ids.record_stream(self._side)
            n_mask.record_stream(self._side)
            ids, mask = self._hop2_inputs(I, D)  # hop-2 inputs of batch i
            ev.append(self._mark())
            q2 = self._encode(ids, mask)
            torch.cuda.current_stream().wait_event(done)
            q_next.record_stream(torch.cuda.current_stream())
            e = torch.cat([q2, q_next], 0) if not self.weak else self._interleave(q2, q_next)
        else:
            ids = mask = None
            ev.append(self._mark())
            q2 = (0.5 * q).repeat_interleave(bm, 0) + self.table[(I.reshape(-1) % 1024)]
            e = torch.cat([q2] + [self.planted_rows + b_["noise"] for b_ in nxt], 0).contiguous()
            if self.weak:
                from .index import all_gather_dim0
                e = all_gather_dim0(e, self.world)
        ev.append(self._mark())
        Dc, Ic = self._search(e.contiguous(), bm)
        if self.weak:
            e, Dc, Ic = self._own(e), self._own(Dc), self._own(Ic)
        q2 = e[:B * bm]
        D2, I2 = Dc[:B * bm].contiguous(), Ic[:B * bm].contiguous()
        for j in range(len(nxt)):
            lo, hi = B * bm + j * B, B * bm + (j + 1) * B
            self._ready.append((e[lo:hi], Dc[lo:hi].contiguous(), Ic[lo:hi].contiguous()))
        ev.append(self._mark())
        h1, h2, sc = rank_paths_device(D, I, D2, I2, bm, self.topk)
        ev.append(self._mark())
        self._ev.append(ev)
        self.step_log.append((self._cur, mask.sum(1) if mask is not None else None))
        self._cur = (self._cur + 1) % self.pool
        return {"q": q, "q2": q2, "D": D, "I": I, "D2": D2, "I2": I2, "hop1": h1, "hop2": h2, "score": sc, "ids2": ids, "mask2": mask}

    def step(self):
        if self.pipelined:
            return self._step_pipelined()
        ev = [self._mark()]
        if self.use_encoder:
            q = self._encode(self.q_ids, self.q_mask)
        else:
            q = self.planted_rows + self.noise
            if self.weak:
                from .index import all_gather_dim0
                q = all_gather_dim0(q, self.world)
        ev.append(self._mark())
        D, I = self._search(q, self.beam)
        if self.weak:  # continue with this rank's own questions
            q, D, I = self._own(q), self._own(D).contiguous(), self._own(I).contiguous()
        ev.append(self._mark())
        if self.use_encoder:
            ids, mask = self._hop2_inputs(I, D)
            ev.append(self._mark())
            q2 = self._encode(ids, mask)
        else:
            ev.append(self._mark())
            q2 = (0.5 * q).repeat_interleave(self.beam, 0) + self.table[(I.reshape(-1) % 1024)]
            if self.weak:
                from .index import all_gather_dim0
                q2 = all_gather_dim0(q2.contiguous(), self.world)
        ev.append(self._mark())
        D2, I2 = self._search(q2.contiguous(), self.beam)
        if self.weak:
            D2, I2 = self._own(D2).contiguous(), self._own(I2).contiguous()
        ev.append(self._mark())
        h1, h2, s = rank_paths_device(D, I, D2, I2, self.beam, self.topk)
        ev.append(self._mark())
        self._ev.append(ev)
        self.last_token_counts = (int(self.Lq), int(self.Lsp))
        self.step_log.append((self._cur, mask.sum(1) if self.use_encoder else None))
        self._cur = (self._cur + 1) % self.pool
        return {"q": q, "q2": q2 if not self.weak else self._own(q2), "D": D, "I": I, "D2": D2, "I2": I2, "hop1": h1, "hop2": h2, "score": s,
                "ids2": ids if self.use_encoder else None, "mask2": mask if self.use_encoder else None}

    # -- reporting -----------------------------------------------------------------------------------------
    def reset_kernel_timers(self):
        self._ev, self._search_ev, self.step_log = [], [], []

    def per_step(self):
        """[(ms, pool entry, hop-2 tokens, hop-2 sequence lengths)] of the timed steps but the last: a step's time runs from its first event to the
        next step's (the main stream is one sequence of steps)."""
        out = []
        for i in range(len(self._ev) - 1):
            ent, lens = self.step_log[i]
            lens = None if lens is None else lens.cpu().numpy()
            out.append((float(self._ev[i][0].elapsed_time(self._ev[i + 1][0])), ent, None if lens is None else int(lens.sum()), lens))
        return out

    def search_kernel_ms(self):
        if not self._search_ev:
            return 0.0
        return float(np.mean([a.elapsed_time(b) for a, b, _ in self._search_ev]))

    def search_calls(self):
        """[(milliseconds, number of queries)] of every timed local search call (HIP events on the launch stream)."""
        return [(float(a.elapsed_time(b)), nq) for a, b, nq in self._search_ev]

    def search_calls_timed(self):
        return len(self._search_ev)

    def stage_ms(self):
        names = ["hop1_encode", "hop1_search", "hop2_assemble", "hop2_encode", "hop2_search", "rank_paths"]
        if not self._ev:
            return {}
        acc = np.zeros(len(names))
        for ev in self._ev:
            acc += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(len(names))])
        return {n: round(float(v / len(self._ev)), 4) for n, v in zip(names, acc)}

    def encoder_desc(self):
        return "hip (RoBERTa-base geometry, random init, fp16 MFMA)" if self.use_encoder else "absent: synthetic query embeddings"

    def self_check(self, out, planted):
        """Cheap structural properties (bench.py adds the full-size exactness check against a brute-force re-scoring of
        the whole corpus): planted hop-1 answers in MIPS-only mode, descending lists, path score = sum of its hop scores,
        every id a valid row."""
        ok = {}
        n = int(self.index.ntotal)
        if not self.use_encoder:
            ok["hop1_top1_is_planted_row"] = bool(torch.equal(out["I"][:, 0], planted))
        if self.beam > 1:
            ok["hop1_sorted"] = bool((out["D"][:, :-1] >= out["D"][:, 1:]).all())
            ok["hop2_sorted"] = bool((out["D2"][:, :-1] >= out["D2"][:, 1:]).all())
        B, bm = self.B, self.beam
        path = (out["D"][:, :, None] + out["D2"].view(B, bm, bm)).view(B, bm * bm)
        best, arg = path.max(1)
        ok["best_path_is_max_of_hop_sums"] = bool(torch.allclose(out["score"][:, 0], best, atol=1e-4)) and bool(torch.isfinite(best).all())
        tie_free = (path == best[:, None]).sum(1) == 1  # the id pair is only defined where the best sum is unique
        i = torch.div(arg, bm, rounding_mode="floor")
        same = (out["hop1"][:, 0] == out["I"][torch.arange(B, device=arg.device), i]) & (out["hop2"][:, 0] == out["I2"].view(B, bm * bm)[torch.arange(B, device=arg.device), arg])
        ok["best_path_ids_match_hop_lists"] = bool((same | ~tie_free).all())
        for name in ("hop1", "hop2", "I", "I2"):
            ok[f"{name}_ids_in_range"] = bool(((out[name] >= 0) & (out[name] < n)).all())
        return ok

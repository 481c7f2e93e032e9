"""Measured-and-rejected variants of the synthetic two-hop loop, kept OUT of the product but runnable:

    python scripts/measure/bench_loops.py --loop deep|shift [--hop1-group G] [--lane-cus N] <bench.py flags>

  deep   (round 4)  two batches in flight: the corpus pass of step i on its own stream beside the encoder forwards of step i+1 -- 14.91 k vs 14.87 k queries/s
  shift  (round 5)  the next hop-1 forward beside the corpus pass instead of beside the hop-2 forward -- +0.7 % (profiles/r05_shift_loop.txt)
  hop1-group G      the questions of G future batches as ONE hop-1 forward every G-th step -- equal (NEGATIVE_RESULTS round 4)
  lane-cus N        CU-partitioned encoder lanes (needs a -DMDR_CU_LANES=1 build of libmdrhip, include/mdr_hip_measure.h) -- slower at every N
                    (profiles/r05_cu_partitioned_lanes_negative.txt)

`VariantTwoHop` subclasses the product's mhop.SyntheticTwoHop; bench_loops.py swaps it in for the main pipeline of bench.main()."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from multihop_dense_retrieval_amd import mhop  # noqa: E402
from multihop_dense_retrieval_amd.mhop import rank_paths_device  # noqa: E402

_Base = mhop.SyntheticTwoHop


class VariantTwoHop(_Base):
    MODE = "pipelined"  # "deep" | "shift" | "pipelined"
    HOP1_GROUP = 1
    LANE_CUS = 0
    _applied = False  # only the FIRST pipelined pipeline bench.main() builds is the variant; its sub-results stay the product loop

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        cls = VariantTwoHop
        self.deep = self.shift = False
        self.hop1_group = 1
        if self.pipelined and not cls._applied:
            cls._applied = True
            self.deep, self.shift = cls.MODE == "deep", cls.MODE == "shift"
            self.hop1_group = max(1, int(cls.HOP1_GROUP))
            self.loop_desc = f"VARIANT loop {cls.MODE}, hop1_group {self.hop1_group}, lane_cus {cls.LANE_CUS} (scripts/measure/loop_variants.py)"
            if self.encoder is not None and cls.LANE_CUS > 0:
                import cu_lanes
                cu_lanes.partition_lanes(self.encoder, cls.LANE_CUS)
            if self.encoder is not None and self.hop1_group > 1:
                self.encoder.capture_on_first_use = True  # a grouped hop-1 shape recurs only every G-th step: capture it at its first sighting
        self._shift_q = None
        self._deep = collections.deque()

    def step(self):
        if self.shift:
            if self.world > 1:
                raise NotImplementedError("--loop shift is a one-rank experiment")
            return self._step_shift()
        if self.deep:
            return self._step_deep()
        if self.pipelined and self.hop1_group > 1:
            return self._step_pipelined_grouped()
        return super().step()

    def _step_pipelined_grouped(self):
        """Batches are independent, so hop 2 of batch i and hop 1 of batch i+1 run as TWO CONCURRENT encoder forwards (two lanes
        = two workspaces and graph caches on two streams, the same weights; one merged 22.6 k-token forward measured slower,
        NEGATIVE_RESULTS.md §6) and share ONE fused corpus pass (B*beam + B queries; more than 128 queries go 256 per pass). Every
        question still walks hop-1 encode -> search -> hop-2 assembly -> hop-2 encode -> search -> path ranking with the same
        arithmetic; what changes is that the small, latency-bound hop-1 forward runs beside the previous batch's large hop-2
        forward and that its queries ride in that batch's corpus pass. In the steady state one call finishes one batch and
        starts the next, i.e. per step exactly one hop-1 and one hop-2 of every kind of work, as in the sequential step.
        `stage_ms()["hop2_encode"]` of this loop is the wall time of BOTH forwards (main stream, waits for the side stream)."""
        G = self.hop1_group
        if not self._ready:
            self._ready.append(self._hop1_only())
            self._search_ev = self._search_ev[:-1] if self._search_ev else self._search_ev  # the prologue is not a timed call
        q, D, I = self._ready.popleft()
        refill = not self._ready  # the next batch's hop 1 is not done yet: this step carries the hop 1 of the next G batches
        B, bm = self.B, self.beam
        nxt = [self.batches[(self._cur + 1 + j) % self.pool] for j in range(G)] if refill else []
        ev = [self._mark()]
        ev.append(ev[0])  # (no separate hop-1 stages)
        ev.append(ev[0])
        q_next = None
        if self.use_encoder:
            done = None
            if refill:
                # the next batches' questions on a side stream / second encoder lane, beside this batch's hop-2 forward: short, latency-bound
                # launches that fill the gaps of the large forward instead of running alone. hop1_group = G > 1 (round 4): the questions of the next
                # G batches as ONE forward every G-th step -- G x the tokens per launch instead of G x the launches
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)  # (a high-priority side stream measured the same: 7.21 vs 7.26 ms)
                start = torch.cuda.Event()
                start.record()
                self._side.wait_event(start)
                n_ids = nxt[0]["q_ids"] if G == 1 else torch.cat([b_["q_ids"] for b_ in nxt], 0)
                n_mask = nxt[0]["q_mask"] if G == 1 else torch.cat([b_["q_mask"] for b_ in nxt], 0)
                with torch.cuda.stream(self._side):
                    q_next = self._encode(n_ids, n_mask, lane=1)
                    done = torch.cuda.Event()
                    done.record()
                n_ids.record_stream(self._side)
                n_mask.record_stream(self._side)
            ids, mask = self._hop2_inputs(I, D)  # hop-2 inputs of batch i
            ev.append(self._mark())
            q2 = self._encode(ids, mask)
            if refill:
                torch.cuda.current_stream().wait_event(done)
                q_next.record_stream(torch.cuda.current_stream())
                e = torch.cat([q2, q_next], 0) if not self.weak else self._interleave(q2, q_next)
            else:
                e = q2
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

    # -- two batches deep: the corpus pass overlaps the next step's encoder forwards ---------------------------------
    def _hop1_of(self, ent):
        b = self.batches[ent]
        if self.use_encoder:
            q = self._encode(b["q_ids"], b["q_mask"])
        else:
            q = self.planted_rows + b["noise"]
            if self.weak:
                from .index import all_gather_dim0
                q = all_gather_dim0(q, self.world)
        D, I = self._search(q, self.beam)
        if self.weak:
            q, D, I = self._own(q), self._own(D).contiguous(), self._own(I).contiguous()
        return q, D, I

    def _step_deep(self):
        """_step_pipelined one batch deeper. There the corpus pass of a step (hop 2 of batch i + hop 1 of batch i+1) stands between the
        step's encoder forwards and the next step's, which need its hop-1 lists: an HBM-bound kernel and an MFMA/L2-bound stage run one
        after the other. Here the pass of step i carries hop 1 of batch i+2, so nothing in step i+1 depends on it: it runs on its OWN
        stream beside the encoder forwards of step i+1 (which wait for the pass of step i-1), and the path ranking of batch i rides behind
        it. Per step every batch still gets one hop-1 forward, one hop-2 forward and its share of one fused corpus pass; only the order
        in which independent batches occupy the GPU changes."""
        B, bm = self.B, self.beam
        if self._search_stream is None:
            self._search_stream = torch.cuda.Stream(device=self.device)
            self._side = torch.cuda.Stream(device=self.device)
        main, side, ss = torch.cuda.current_stream(), self._side, self._search_stream
        if not self._deep:  # prologue: hop 1 of the first two batches, plainly
            for k in range(2):
                q, D, I = self._hop1_of((self._cur + k) % self.pool)
                e = torch.cuda.Event()
                e.record()
                self._deep.append((q, D, I, e))
            self._search_ev = []
        q, D, I, ready = self._deep.popleft()
        main.wait_event(ready)  # the pass that carried this batch's hop 1 (issued two steps ago)
        for t in (q, D, I):
            t.record_stream(main)
        ev = [self._mark()]
        ev.append(ev[0])
        ev.append(ev[0])
        nb = self.batches[(self._cur + 2) % self.pool]
        if self.use_encoder:
            start = torch.cuda.Event()
            start.record()
            side.wait_event(start)
            with torch.cuda.stream(side):
                q_next = self._encode(nb["q_ids"], nb["q_mask"], lane=1)
                done = torch.cuda.Event()
                done.record()
            ids, mask = self._hop2_inputs(I, D)
            ev.append(self._mark())
            q2 = self._encode(ids, mask)
        else:
            ids = mask = None
            ev.append(self._mark())
            q2 = (0.5 * q).repeat_interleave(bm, 0) + self.table[(I.reshape(-1) % 1024)]
            q_next = self.planted_rows + nb["noise"]
            if self.weak:
                from .index import all_gather_dim0
                q2, q_next = all_gather_dim0(q2.contiguous(), self.world), all_gather_dim0(q_next, self.world)
            done = None
        ev.append(self._mark())
        enc_done = torch.cuda.Event()
        enc_done.record()
        ss.wait_event(enc_done)
        if done is not None:
            ss.wait_event(done)
        with torch.cuda.stream(ss):
            for t in (q2, q_next, D, I):
                t.record_stream(ss)
            e = torch.cat([q2, q_next], 0) if not self.weak else self._interleave(q2, q_next)
            Dc, Ic = self._search(e.contiguous(), bm)
            if self.weak:
                e, Dc, Ic = self._own(e), self._own(Dc), self._own(Ic)
            q2o, qn = e[:B * bm], e[B * bm:]
            D2, I2 = Dc[:B * bm].contiguous(), Ic[:B * bm].contiguous()
            Dn, In = Dc[B * bm:].contiguous(), Ic[B * bm:].contiguous()
            ev.append(self._mark())
            h1, h2, sc = rank_paths_device(D, I, D2, I2, bm, self.topk)
            ev.append(self._mark())
            fin = torch.cuda.Event()
            fin.record()
        self._deep.append((qn, Dn, In, fin))
        self._ev.append(ev)  # (stages overlap across steps here: hop2_search / rank_paths run beside the NEXT step's encoder stage)
        self.step_log.append((self._cur, mask.sum(1) if mask is not None else None))
        self._cur = (self._cur + 1) % self.pool
        return {"q": q, "q2": q2o, "D": D, "I": I, "D2": D2, "I2": I2, "hop1": h1, "hop2": h2, "score": sc, "ids2": ids, "mask2": mask}

    # -- hop 1 beside the corpus pass ------------------------------------------------------------------------------------------------
    def _step_shift(self):
        """_step_pipelined with the small forward moved: there the next batch's hop-1 forward (~84 short kernels) runs beside the hop-2 forward, whose
        persistent one-workgroup-per-CU GEMMs own every CU -- each short kernel waits for one of them to end and delays the next one by its own duration
        (0.74 ms of a 5.4 ms encoder stage). Here the hop-2 forward of batch i runs ALONE; the hop-1 forward of batch i+2 starts when it ends, on the
        side stream, beside the fused corpus pass of step i (hop 2 of batch i + hop 1 of batch i+1, whose embeddings were produced beside the pass of
        step i-1). An HBM / int8-MFMA-bound pass and a latency-bound small forward share the chip instead of two fp16-MFMA forwards. What spills over the
        end of the pass overlaps the next hop-2 forward as before. Same arithmetic per question; only the order of independent batches changes."""
        B, bm = self.B, self.beam
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main, side = torch.cuda.current_stream(), self._side
        if not self._ready:  # prologue: hop 1 of the first batch plainly; the second batch's questions encoded, to be searched in the first fused pass
            self._ready.append(self._hop1_only())
            nb = self._nxt()
            qn = self._encode(nb["q_ids"], nb["q_mask"], lane=1) if self.use_encoder else self.planted_rows + nb["noise"]
            e0 = torch.cuda.Event()
            e0.record()
            self._shift_q = (qn, e0)
            self._search_ev = self._search_ev[:-1] if self._search_ev else self._search_ev
        q, D, I = self._ready.popleft()
        ev = [self._mark()]
        ev.append(ev[0])
        ev.append(ev[0])
        nb2 = self.batches[(self._cur + 2) % self.pool]
        if self.use_encoder:
            ids, mask = self._hop2_inputs(I, D)
            ev.append(self._mark())
            q2 = self._encode(ids, mask)
            enc_done = torch.cuda.Event()
            enc_done.record()
            side.wait_event(enc_done)  # the small forward starts when the large one has ended
            with torch.cuda.stream(side):
                q_nn = self._encode(nb2["q_ids"], nb2["q_mask"], lane=1)
                done = torch.cuda.Event()
                done.record()
            nb2["q_ids"].record_stream(side)
            nb2["q_mask"].record_stream(side)
        else:
            ids = mask = None
            ev.append(self._mark())
            q2 = (0.5 * q).repeat_interleave(bm, 0) + self.table[(I.reshape(-1) % 1024)]
            q_nn, done = self.planted_rows + nb2["noise"], None
        ev.append(self._mark())
        q_next, ready_ev = self._shift_q
        main.wait_event(ready_ev)
        q_next.record_stream(main)
        e = torch.cat([q2, q_next], 0)
        Dc, Ic = self._search(e.contiguous(), bm)
        q2 = e[:B * bm]
        D2, I2 = Dc[:B * bm].contiguous(), Ic[:B * bm].contiguous()
        self._ready.append((e[B * bm:], Dc[B * bm:].contiguous(), Ic[B * bm:].contiguous()))
        ev.append(self._mark())
        h1, h2, sc = rank_paths_device(D, I, D2, I2, bm, self.topk)
        ev.append(self._mark())
        if done is None:
            done = torch.cuda.Event()
            done.record()
        self._shift_q = (q_nn, done)
        self._ev.append(ev)
        self.step_log.append((self._cur, mask.sum(1) if mask is not None else None))
        self._cur = (self._cur + 1) % self.pool
        return {"q": q, "q2": q2, "D": D, "I": I, "D2": D2, "I2": I2, "hop1": h1, "hop2": h2, "score": sc, "ids2": ids, "mask2": mask}


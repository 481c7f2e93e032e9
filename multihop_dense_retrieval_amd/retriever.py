"""Host-side mirror of the reference's encoder wrappers, backed by libmdrhip.so:

    RobertaRetriever(config, args).encode_q(input_ids, mask, type_ids) -> [B, hidden]
        /root/reference/mdr/retrieval/models/mhop_retriever.py:12-41
    RobertaCtxEncoder(config, args)(batch)['embed']
        /root/reference/mdr/retrieval/models/retriever.py:176-190
    load_saved(model, path, exact=True)
        /root/reference/mdr/retrieval/utils/utils.py:10-22
    move_to_cuda(sample)
        /root/reference/mdr/retrieval/utils/utils.py:24-41

Same names, argument meaning and error behaviour, so the body of the reference's eval loop ports
line for line. The arithmetic is not here: `load_state_dict` hands the fp32 tensors to
mdr_encoder_create (csrc/mdr_encoder.hip) and `encode_seq` calls mdr_encoder_forward.
"""
import ctypes
import os
from collections import OrderedDict

import torch

from . import _lib


class RobertaConfig:
    """The geometry constants of roberta-base. An HF config object works too:
    only these attribute names are read."""

    def __init__(self, vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=514, layer_norm_eps=1e-5, pad_token_id=1, **_):
        self.vocab_size, self.hidden_size, self.num_hidden_layers = vocab_size, hidden_size, num_hidden_layers
        self.num_attention_heads, self.intermediate_size = num_attention_heads, intermediate_size
        self.max_position_embeddings, self.layer_norm_eps, self.pad_token_id = max_position_embeddings, layer_norm_eps, pad_token_id


def expected_state_dict_shapes(config, with_pooler=True):
    """Keys and shapes of the reference model's state_dict (HF RobertaModel + project head)."""
    H, F = config.hidden_size, config.intermediate_size
    sd = OrderedDict()
    e = "encoder.embeddings."
    sd[e + "word_embeddings.weight"] = (config.vocab_size, H)
    sd[e + "position_embeddings.weight"] = (config.max_position_embeddings, H)
    sd[e + "token_type_embeddings.weight"] = (1, H)
    sd[e + "LayerNorm.weight"] = (H,)
    sd[e + "LayerNorm.bias"] = (H,)
    for i in range(config.num_hidden_layers):
        p = f"encoder.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            sd[p + f"attention.self.{n}.weight"] = (H, H)
            sd[p + f"attention.self.{n}.bias"] = (H,)
        sd[p + "attention.output.dense.weight"] = (H, H)
        sd[p + "attention.output.dense.bias"] = (H,)
        sd[p + "attention.output.LayerNorm.weight"] = (H,)
        sd[p + "attention.output.LayerNorm.bias"] = (H,)
        sd[p + "intermediate.dense.weight"] = (F, H)
        sd[p + "intermediate.dense.bias"] = (F,)
        sd[p + "output.dense.weight"] = (H, F)
        sd[p + "output.dense.bias"] = (H,)
        sd[p + "output.LayerNorm.weight"] = (H,)
        sd[p + "output.LayerNorm.bias"] = (H,)
    if with_pooler:
        sd["encoder.pooler.dense.weight"] = (H, H)
        sd["encoder.pooler.dense.bias"] = (H,)
    sd["project.0.weight"] = (H, H)
    sd["project.0.bias"] = (H,)
    sd["project.1.weight"] = (H,)
    sd["project.1.bias"] = (H,)
    return sd


class _Lane:
    def __init__(self):
        self.ws = None
        self.graphs = OrderedDict()  # (B, L bucket) -> capture, LRU order
        self.seen = {}


class _HipRobertaEncoder:
    """Shared implementation of the two reference classes (they compute the same function,
    SURVEY.md §8a row a23)."""

    MAX_TOKENS_PER_CALL = 1 << 17  # workspace bound: larger batches are encoded in slices
    # numerics mode of the residual stream (mdr_encoder_config.residual_fp32): 0 = fp16 residual copy (one rounding more per LayerNorm than apex O1),
    # 1 = fp32 residual stream + fp32 Linear sums, 2 = fp32 residual stream + out-projection / FFN2 outputs rounded to fp16 (literally apex O1's dataflow
    # around the LayerNorms, the reference's regime: eval_mhop_retrieval.py:86-90). 1 and 2 are the O1-faithful modes; 2 moves 4 bytes per element less.
    RESIDUAL_FP32_DEFAULT = 2

    def __init__(self, config, args=None):
        self.config = config
        self.args = args
        self._shapes = expected_state_dict_shapes(config)
        self._h = ctypes.c_void_p()
        # per-lane scratch state: a lane = one workspace + its captured graphs. Two forwards may be in flight at once (on two
        # streams) when they use different lanes; the C handle itself only holds the weights.
        self._lanes = {}
        self.capture_on_first_use = False
        self.graph_captures = 0
        self.graph_replays = 0
        self.forward_calls = 0  # encode_seq invocations / sequences encoded (the multi-rank tests assert each rank's share)
        self.forward_rows = 0
        self.use_graphs = True
        # apex-O1-faithful fp32 residual stream (mdr_encoder_config.residual_fp32); MDR_RESIDUAL_FP32=0/1/2 overrides the default for
        # measurements before the weights are uploaded
        self.residual_fp32 = int(os.environ.get("MDR_RESIDUAL_FP32", str(int(self.RESIDUAL_FP32_DEFAULT))))
        self.device = None
        self.training = False

    # -- nn.Module-like surface used by the reference scripts -------------------------------------------
    def state_dict(self):
        """Key set only (values are shapes): load_saved(exact=False) filters a checkpoint with it."""
        return self._shapes

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._shapes if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: Missing key(s): {missing}. Unexpected key(s): {unexpected}.")
        for k, shp in self._shapes.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(state_dict[k].shape)} vs model {tuple(shp)}")
        self._pending = {k: v for k, v in state_dict.items() if k in self._shapes}
        if self.device is not None:
            self._create()
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the encoder runs on a HIP device only (there is no CPU fallback)")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        if getattr(self, "_pending", None) is not None:
            self._create()
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        self.training = False
        return self

    def half(self):  # apex-O1-equivalent numerics are built in
        return self

    # -- the forward ----------------------------------------------------------------------------------------
    GRAPH_L_BUCKET = 32       # graph keys use seq_len rounded up to this (padding columns are free: execution is un-padded)
    GRAPH_CACHE_ENTRIES = 24  # LRU bound on captured shapes

    def _lane(self, lane):
        st = self._lanes.get(lane)
        if st is None:
            st = self._lanes[lane] = _Lane()
        return st

    def bind_lane_stream(self, lane, stream):
        """Bind a lane to a stream of its own: the lane's forwards then run (and are captured) on that stream whatever stream the caller is on, ordered with the
        caller's stream by events. None removes the binding. Captured graphs of every lane are dropped (a capture freezes the stream it was made on)."""
        if not hasattr(self, "_lane_streams"):
            self._lane_streams = {}
        for st in self._lanes.values():
            st.graphs.clear()
        if stream is None:
            self._lane_streams.pop(lane, None)
        else:
            self._lane_streams[lane] = stream

    def lane_stream(self, lane):
        return getattr(self, "_lane_streams", {}).get(lane)

    def encode_seq(self, input_ids, mask, lane=0):
        """lane: which workspace / graph cache this call uses. Calls on DIFFERENT lanes may overlap on different streams (the
        pipelined retrieval loop encodes the next batch's questions beside the current batch's hop-2 inputs)."""
        ls = self.lane_stream(lane)
        if ls is not None and torch.cuda.current_stream(self.device) != ls:
            # this lane is bound to its own stream: run there, ordered behind what the caller's stream holds, and let the caller's stream wait for the result
            cur = torch.cuda.current_stream(self.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            ls.wait_event(ev)
            with torch.cuda.stream(ls):
                out = self.encode_seq(input_ids, mask, lane)
                done = torch.cuda.Event()
                done.record(ls)
            for t in (input_ids, mask):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(ls)
            cur.wait_event(done)
            out.record_stream(cur)
            return out
        if not self._h.value:
            raise RuntimeError("encoder has no weights on a device: call load_saved(...)/load_state_dict(...) and .to('cuda') first")
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        msk = mask.to(device=self.device, dtype=torch.int64).contiguous()
        if ids.dim() != 2 or ids.shape != msk.shape:
            raise ValueError(f"input_ids {tuple(ids.shape)} and mask {tuple(msk.shape)} must both be [B, L]")
        B, L = ids.shape
        self.forward_calls += 1
        self.forward_rows += int(B)
        if self.use_graphs and 0 < B * L <= self.MAX_TOKENS_PER_CALL and L <= 512:
            return self._encode_graphed(ids, msk, lane)
        out = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=self.device)
        self._forward_into(ids, msk, out, lane)
        return out

    def _forward_into(self, ids, msk, out, lane=0):
        st = self._lane(lane)
        B, L = ids.shape
        per = max(1, self.MAX_TOKENS_PER_CALL // L)
        L_ = _lib.lib()
        for lo in range(0, B, per):
            hi = min(B, lo + per)
            need = int(L_.mdr_encoder_workspace_bytes(self._h, hi - lo, L))
            if st.ws is None or st.ws.numel() < need:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("encoder workspace must be sized before graph capture")
                st.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.check(L_.mdr_encoder_forward(self._h, ctypes.c_void_p(ids[lo:hi].data_ptr()), ctypes.c_void_p(msk[lo:hi].data_ptr()), hi - lo, L,
                                              ctypes.c_void_p(out[lo:hi].data_ptr()), ctypes.c_void_p(st.ws.data_ptr()), st.ws.numel(),
                                              _lib.current_stream_ptr(self.device)))

    def _encode_graphed(self, ids, msk, lane=0):
        """One forward is ~140 short kernel launches; for the small batches of the retrieval loop (and for each
        rank's slice under multi-GPU data parallelism) the launch gaps dominate. The launch sequence depends only
        on (B, L), so it is captured into a hipGraph and replayed (static input/output buffers).

        Shapes are keyed by (B, L rounded up to GRAPH_L_BUCKET): corpus encoding pads every batch to its own longest
        passage (em_collate), which would otherwise make nearly every batch a new shape. The extra columns are pad/mask-0
        and cost nothing (masked tokens are dropped on the device before any arithmetic). A shape is captured the SECOND
        time it is seen (a one-off shape runs eagerly, without the warm-up + capture cost); the cache is LRU-bounded and
        is dropped as a whole when the workspace is re-allocated (captures hold raw pointers into it)."""
        B, L = ids.shape
        Lb = min(512, -(-L // self.GRAPH_L_BUCKET) * self.GRAPH_L_BUCKET)
        st = self._lane(lane)
        key = (B, Lb)
        ent = st.graphs.get(key)
        if ent is not None and ent[4] is not st.ws:  # stale: the workspace moved since the capture
            st.graphs.clear()
            ent = None
        if ent is None:
            seen = st.seen.get(key, 0) + 1
            st.seen[key] = seen
            if seen < 2 and not self.capture_on_first_use:
                out = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=self.device)
                self._forward_into(ids, msk, out, lane)
                return out
            sid = torch.full((B, Lb), int(self.config.pad_token_id), dtype=torch.int64, device=self.device)
            smk = torch.zeros((B, Lb), dtype=torch.int64, device=self.device)
            sid[:, :L].copy_(ids)
            smk[:, :L].copy_(msk)
            sout = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=self.device)
            # the capture freezes the tile-shape choices: tell the library how full this shape's batches are (the first
            # batch stands for the later ones of the same shape; only speed depends on it). One sync, at capture time only.
            fill = float(smk.sum().item()) / float(smk.numel()) if os.environ.get("MDR_FILL_HINT", "1") != "0" else 0.0
            _lib.check(_lib.lib().mdr_encoder_set_fill_hint(self._h, min(1.0, max(fill, 1e-3)) if fill > 0 else 0.0))
            try:
                self._forward_into(sid, smk, sout, lane)  # warm-up: sizes the workspace, sets kernel attributes
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                ls = self.lane_stream(lane)  # a lane bound to its own stream is captured ON it
                with (torch.cuda.graph(graph, stream=ls) if ls is not None else torch.cuda.graph(graph)):
                    self._forward_into(sid, smk, sout, lane)
            finally:
                _lib.check(_lib.lib().mdr_encoder_set_fill_hint(self._h, 0.0))
            ent = (graph, sid, smk, sout, st.ws)
            while len(st.graphs) >= self.GRAPH_CACHE_ENTRIES:
                st.graphs.popitem(last=False)  # least recently used
            st.graphs[key] = ent
            self.graph_captures += 1
            return sout.clone()  # the warm-up / capture pair already produced this batch's result
        st.graphs.move_to_end(key)
        graph, sid, smk, sout, _ = ent
        sid[:, :L].copy_(ids)
        smk[:, :L].copy_(msk)
        if L < Lb:
            smk[:, L:].zero_()
        graph.replay()
        self.graph_replays += 1
        return sout.clone()

    def precapture(self, batch, seq_len, fill=0.5, lane=0):
        """Capture the hipGraph of one (batch, seq_len) shape on `lane` ahead of time, on a synthetic batch whose rows hold fill * seq_len tokens (the
        fraction only steers tile-shape choices, mdr_encoder_set_fill_hint). The drop-in CLI calls it while it loads the model: its loop repeats two
        shapes, and capturing them inside the loop costs the first batch ~70 ms."""
        if not self.use_graphs or batch <= 0 or batch * seq_len > self.MAX_TOKENS_PER_CALL or seq_len > 512:
            return
        n = max(2, min(seq_len, int(round(fill * seq_len))))
        ids = torch.full((batch, seq_len), int(self.config.pad_token_id), dtype=torch.int64, device=self.device)
        ids[:, :n] = 5
        ids[:, 0] = 0
        ids[:, n - 1] = 2
        mask = torch.zeros((batch, seq_len), dtype=torch.int64, device=self.device)
        mask[:, :n] = 1
        prev, calls, rows = self.capture_on_first_use, self.forward_calls, self.forward_rows
        self.capture_on_first_use = True
        try:
            self.encode_seq(ids, mask, lane)
        finally:
            self.capture_on_first_use = prev
            self.forward_calls, self.forward_rows = calls, rows  # (not a forward of the caller's data)

    # -- internals ----------------------------------------------------------------------------------------------
    def _create(self):
        sd = self._pending
        cfg = _lib.EncoderConfig(self.config.vocab_size, self.config.hidden_size, self.config.num_hidden_layers, self.config.num_attention_heads,
                                 self.config.intermediate_size, self.config.max_position_embeddings, self.config.pad_token_id,
                                 float(self.config.layer_norm_eps), int(self.residual_fp32))
        names = [k for k in sd if "pooler" not in k]  # the pooler is never evaluated (`[0]` = sequence output)
        on_dev = all(sd[k].is_cuda for k in names)
        keep = []
        arr = (_lib.Tensor * len(names))()
        for i, k in enumerate(names):
            t = sd[k].detach().to(dtype=torch.float32)
            t = t.contiguous() if on_dev else t.cpu().contiguous()
            keep.append(t)
            arr[i] = _lib.Tensor(k.encode(), ctypes.c_void_p(t.data_ptr()), t.numel())
        if self._h.value:
            _lib.lib().mdr_encoder_free(self._h)
            self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mdr_encoder_create(ctypes.byref(cfg), arr, len(names), int(on_dev), self.device.index,
                                                     _lib.current_stream_ptr(self.device), ctypes.byref(self._h)))
        self._pending = None

    def __del__(self):
        try:
            if self._h.value:
                _lib.lib().mdr_encoder_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    @classmethod
    def random_init(cls, device, seed=0, config=None):
        """Random weights of the right geometry (bench / smoke: no checkpoints offline)."""
        config = config or RobertaConfig()
        m = cls(config, None)
        g = torch.Generator(device=device).manual_seed(seed)
        sd = {}
        for k, shp in m._shapes.items():
            if k.endswith("LayerNorm.weight") or k == "project.1.weight":
                sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device)
            elif k.endswith("bias"):
                sd[k] = 0.1 * torch.randn(shp, generator=g, device=device)
            elif "embeddings" in k:
                sd[k] = 0.5 * torch.randn(shp, generator=g, device=device)
            else:
                sd[k] = (1.5 / shp[1] ** 0.5) * torch.randn(shp, generator=g, device=device)
        m.load_state_dict(sd)
        return m.to(device).eval()


class RobertaRetriever(_HipRobertaEncoder):
    """mhop_retriever.py:12-41 -- query encoder; `encode_q(ids, mask, type_ids)` ignores type_ids like the reference."""

    def encode_q(self, input_ids, q_mask, q_type_ids=None, lane=0):
        return self.encode_seq(input_ids, q_mask, lane)

    def __call__(self, batch):
        raise NotImplementedError("training forward (six encode_seq calls, mhop_retriever.py:28-38) is outside the retrieval hot path")


class RobertaCtxEncoder(_HipRobertaEncoder):
    """retriever.py:176-190 -- passage encoder; `model(batch)` with keys input_ids / input_mask -> {'embed': ...}."""

    def forward(self, batch):
        return {"embed": self.encode_seq(batch["input_ids"], batch["input_mask"])}

    __call__ = forward


def load_saved(model, path, exact=True, map_location=None):
    """utils.py:10-22 -- load a (possibly `module.`-prefixed) state dict; exact=False drops keys the model
    does not have, then loads strictly (so a MISSING key still raises). map_location (not in the reference): "cpu" keeps a checkpoint that was
    saved from GPU tensors in host memory (the drop-in CLI loads it before it touches the device; the weights are uploaded by model.to())."""
    try:
        state_dict = torch.load(path) if map_location is None else torch.load(path, map_location=map_location)
    except Exception:
        state_dict = torch.load(path, map_location=torch.device("cpu"))

    def strip(k):
        return k[7:] if k.startswith("module.") else k

    known = model.state_dict()
    if exact:
        state_dict = {strip(k): v for k, v in state_dict.items()}
    else:
        state_dict = {strip(k): v for k, v in state_dict.items() if strip(k) in known}
    model.load_state_dict(state_dict)
    return model


def move_to_cuda(sample):
    """utils.py:24-41 -- recursive .cuda() over dicts / lists of tensors."""
    if len(sample) == 0:
        return {}

    def mv(x):
        if torch.is_tensor(x):
            return x.cuda()
        if isinstance(x, dict):
            return {k: mv(v) for k, v in x.items()}
        if isinstance(x, list):
            return [mv(v) for v in x]
        return x

    return mv(sample)


"""Flat inner-product index on one MI355X -- the host-side mirror of `faiss.IndexFlatIP` as the
reference uses it (/root/reference/scripts/eval/eval_mhop_retrieval.py:121-125,155,179):

    index = IndexFlatIP(d); index.add(xb); D, I = index.search(x, k)

Same names, argument meaning and result layout (D float32 [n,k] descending, I int64 [n,k], -1 /
-FLT_MAX padding when fewer than k rows). numpy in -> numpy out (drop-in for the reference loop);
torch CUDA tensor in -> torch CUDA tensors out with no host round trip and no device sync.

All arithmetic happens in libmdrhip.so (csrc/mdr_mips.hip); this file only moves pointers.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_TORCH_DT = {torch.float32: _lib.MDR_DT_F32, torch.bfloat16: _lib.MDR_DT_BF16, torch.float16: _lib.MDR_DT_F16}
_NP_DT = {np.dtype(np.float32): _lib.MDR_DT_F32, np.dtype(np.float16): _lib.MDR_DT_F16}


class IndexFlatIP:
    """Brute-force maximum-inner-product index resident in HBM."""

    def __init__(self, d, device=None, storage=_lib.MDR_STORE_F32X2H):
        """storage: MDR_STORE_F32X2H (default; fp32-accurate, 4 B/element + at d = 768 a 1 B/element int8 screening copy that makes k = 1
        searches ~1.5 x faster), MDR_STORE_F32X2H_COMPACT / "compact" (the same without that copy: faiss.IndexFlatIP's 4 B/element, same
        results) or MDR_STORE_BF16 / "bf16" (rows rounded to bf16, 2 B/element; scores are exact w.r.t. the rounded rows, i.e. within 1e-2 of
        the fp32 scores for unit-scale data)."""
        if storage in ("bf16", "BF16"):
            storage = _lib.MDR_STORE_BF16
        elif storage in ("compact", "f32-compact"):
            storage = _lib.MDR_STORE_F32X2H_COMPACT
        if not torch.cuda.is_available():
            raise RuntimeError("IndexFlatIP needs a HIP device (there is no CPU fallback)")
        self.d = int(d)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().mdr_index_create(self.d, storage, self.device.index, ctypes.byref(self._h)))
        self._ws = None
        self.id_offset = 0  # global id of local row 0 (set by ShardedIndexFlatIP)

    # -- faiss-compatible surface ---------------------------------------------------------------------
    @property
    def ntotal(self):
        return int(_lib.lib().mdr_index_ntotal(self._h))

    def reserve(self, n_rows):
        _lib.check(_lib.lib().mdr_index_reserve(self._h, int(n_rows)))

    def add(self, x):
        """Append rows. x: [n, d] numpy (float32/float16) or torch tensor (cpu or cuda; f32/f16/bf16)."""
        L = _lib.lib()
        if isinstance(x, np.ndarray):
            if x.dtype not in _NP_DT:
                x = x.astype(np.float32)  # eval_mhop_retrieval.py:94 `.astype('float32')`
            x = np.ascontiguousarray(x)
            self._check_shape(x.shape)
            # big host arrays (np.load(mmap_mode='r') works too) go through the library's chunked upload
            _lib.check(L.mdr_index_add(self._h, ctypes.c_void_p(x.ctypes.data), x.shape[0], _NP_DT[x.dtype], 0,
                                       _lib.current_stream_ptr(self.device)))
            return
        if not torch.is_tensor(x):
            raise TypeError("add() expects a numpy array or a torch tensor")
        if x.dtype not in _TORCH_DT:
            x = x.float()
        x = x.contiguous()
        self._check_shape(tuple(x.shape))
        if x.is_cuda and x.device != self.device:
            x = x.to(self.device)
        _lib.check(L.mdr_index_add(self._h, ctypes.c_void_p(x.data_ptr()), x.shape[0], _TORCH_DT[x.dtype], int(x.is_cuda),
                                   _lib.current_stream_ptr(self.device)))

    def search(self, x, k):
        """D, I = top-k rows by inner product, best first. numpy in -> numpy out; cuda tensor in -> cuda out."""
        as_numpy = isinstance(x, np.ndarray)
        if as_numpy:
            q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
        else:
            q = x.to(device=self.device, dtype=torch.float32).contiguous()
        D, I = self.search_device(q, k)
        if as_numpy:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    def search_device(self, q, k, out=None):
        """q: float32 cuda [nq, d] contiguous. Enqueues on torch's current stream; no sync."""
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"query shape {tuple(q.shape)} does not match index dimension {self.d}")
        if q.dtype != torch.float32 or not q.is_cuda or not q.is_contiguous():
            raise ValueError("search_device() wants a contiguous float32 CUDA tensor")
        nq, k = int(q.shape[0]), int(k)
        L = _lib.lib()
        need = int(L.mdr_index_search_workspace_bytes(self._h, nq, k))
        if need == 0 and nq > 0:
            raise ValueError(f"k={k} out of range")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            D, I = out
        _lib.check(L.mdr_index_search(self._h, ctypes.c_void_p(q.data_ptr()), nq, k, ctypes.c_void_p(D.data_ptr()),
                                      ctypes.c_void_p(I.data_ptr()), int(self.id_offset), ctypes.c_void_p(self._ws.data_ptr()),
                                      self._ws.numel(), _lib.current_stream_ptr(self.device)))
        return D, I

    def search_device_packed(self, q, k):
        """search_device() writing (D, I) straight into ONE packed block (mdr_topk_packed_bytes: scores at offset 0, ids at
        mdr_topk_packed_ids_offset) -- the unit the sharded search all-gathers, so the exchange needs no glue kernels.
        -> (block uint8 [bytes], D view float32 [nq, k], I view int64 [nq, k])."""
        L = _lib.lib()
        nq, k = int(q.shape[0]), int(k)
        nbytes, off = int(L.mdr_topk_packed_bytes(nq, k)), int(L.mdr_topk_packed_ids_offset(nq, k))
        block = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=self.device)
        D = block[: nq * k * 4].view(torch.float32).view(nq, k)
        I = block[off: off + nq * k * 8].view(torch.int64).view(nq, k)
        self.search_device(q, k, out=(D, I))
        return block, D, I

    # -- extras ------------------------------------------------------------------------------------------
    def stream_bytes(self):
        """HBM bytes one search call reads for the corpus (algorithmic bytes of the roofline)."""
        return int(_lib.lib().mdr_index_stream_bytes(self._h))

    def queries_per_pass(self, nq, k=1):
        """Queries one pass over the shard serves for a call of nq queries (the corpus is read ceil(nq / this) times)."""
        return int(_lib.lib().mdr_index_queries_per_pass(self._h, int(nq), int(k)))

    def telemetry(self, nq, k):
        """Test hook (synchronises): {"fallback", "candidates", "bad_query", "path", "i8_tier", "i8_overflow", "i8_query_split", "i8_refined"} of the last search of this shape."""
        out = (ctypes.c_int64 * 4)()
        _lib.check(_lib.lib().mdr_index_search_telemetry(self._h, int(nq), int(k), ctypes.c_void_p(self._ws.data_ptr()), out,
                                                         _lib.current_stream_ptr(self.device)))
        return {"fallback": int(out[0]), "candidates": int(out[1]), "bad_query": int(out[2]), "path": int(out[3]) & 0xFF,
                "i8_tier": bool(int(out[3]) & 512), "i8_overflow": bool(int(out[3]) & 256), "i8_query_split": bool(int(out[3]) & 1024),
                "i8_refined": int(out[3]) >> 16}

    def set_variant(self, v):
        """Test hook: 0 auto, 1 generic fp32 kernel, 2 exact 3-MFMA stream kernel, 3 screen + refine, 4 screen + refine without the int8 tier."""
        _lib.check(_lib.lib().mdr_index_set_variant(self._h, int(v)))

    def last_kernel(self):
        return _lib.lib().mdr_index_last_kernel(self._h).decode()

    def _check_shape(self, shape):
        if len(shape) != 2 or shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] rows, got {shape}")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                _lib.lib().mdr_index_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


def topk_merge(D_parts, I_parts):
    """[P, nq, k] cuda float32 / int64 -> merged (D [nq,k], I [nq,k]); score desc, id asc."""
    P, nq, k = D_parts.shape
    D_parts = D_parts.contiguous()
    I_parts = I_parts.contiguous()
    D = torch.empty((nq, k), dtype=torch.float32, device=D_parts.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_parts.device)
    _lib.check(_lib.lib().mdr_topk_merge(ctypes.c_void_p(D_parts.data_ptr()), ctypes.c_void_p(I_parts.data_ptr()), P, nq, k,
                                         ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                         _lib.current_stream_ptr(D_parts.device)))
    return D, I


def topk_merge_packed(blocks, nparts, nq, k):
    """blocks: uint8 cuda [nparts * mdr_topk_packed_bytes(nq, k)] (the all-gathered search_device_packed() blocks of the ranks)
    -> merged (D [nq, k], I [nq, k]); ONE kernel, no unpacking (mdr_topk_merge_packed)."""
    D = torch.empty((nq, k), dtype=torch.float32, device=blocks.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=blocks.device)
    _lib.check(_lib.lib().mdr_topk_merge_packed(ctypes.c_void_p(blocks.data_ptr()), int(nparts), int(nq), int(k), ctypes.c_void_p(D.data_ptr()),
                                                ctypes.c_void_p(I.data_ptr()), _lib.current_stream_ptr(blocks.device)))
    return D, I


def all_gather_dim0(t, world, group=None):
    """all_gather_into_tensor along dim 0. RCCL ("nccl") takes device tensors directly; with the gloo backend
    (CPU tests, or several ranks sharing one GPU while debugging) device tensors are staged through the host."""
    import torch.distributed as dist
    out_shape = (world * t.shape[0],) + tuple(t.shape[1:])
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        out = torch.empty(out_shape, dtype=t.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(t.device)
    out = torch.empty(out_shape, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def shard_bounds(n_total, world_size, rank):
    """Contiguous row blocks: rank r owns [r*ceil(N/W), min(N, (r+1)*ceil(N/W)))  (SURVEY.md §8e)."""
    per = -(-int(n_total) // int(world_size))
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


class ShardedIndexFlatIP:
    """Row-sharded flat index over the ranks of a torch.distributed process group (one process per GPU).

    Each rank searches its own shard, the per-shard (D, I) lists are exchanged with ONE all_gather per
    search (RCCL over xGMI; KB-sized, latency-bound) and every rank runs the same deterministic merge,
    so all ranks hold identical hop-1 results and can build hop-2 queries without more communication.

    `local_index` / `merge_fn` are injectable so the world_size-2 gloo tests on CPU can exercise the
    partitioning, the collective and the id arithmetic with the oracle standing in for the kernels.
    """

    def __init__(self, d, n_total, group=None, local_index=None, merge_fn=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.d = d
        self.n_total = int(n_total)
        self.lo, self.hi = shard_bounds(self.n_total, self.world, self.rank)
        self.local = local_index if local_index is not None else IndexFlatIP(d)
        self.local.id_offset = self.lo
        self.merge_fn = merge_fn if merge_fn is not None else topk_merge

    @property
    def ntotal(self):
        return self.n_total

    def add_local(self, x_local):
        """Rows [lo, hi) of the global matrix, in order."""
        self.local.add(x_local)

    def add_from_global(self, xb):
        """Convenience: slice this rank's rows out of a full (e.g. mmap'ed) matrix."""
        self.local.add(xb[self.lo:self.hi])

    def search(self, q, k, force=False):
        """faiss-style surface over the shards: numpy in -> numpy out, cuda tensor in -> cuda tensors out. With the real
        local index the exchange and the merge always run on DEVICE tensors (RCCL takes device buffers and mdr_topk_merge
        is a HIP kernel): numpy queries are uploaded once and only the merged result comes back to the host."""
        as_numpy = isinstance(q, np.ndarray)
        if as_numpy and isinstance(self.local, IndexFlatIP):
            qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(self.local.device)
            D, I = self.local.search_device(qd, k)
            Dm, Im = self.search_gathered(D, I, force=force)
            return Dm.cpu().numpy(), Im.cpu().numpy()
        D, I = self.local.search(q, k)
        return self.search_gathered(D, I, force=force)

    def search_device(self, q, k, force=False):
        """Device tensors in, device tensors out, no sync: this rank's shard is searched with the results written straight into the
        packed exchange block, ONE all-gather (RCCL over xGMI) moves every rank's block, ONE kernel merges them (mdr_topk_merge_packed)
        -- besides the search itself the hop costs one collective and one launch. Identical on all ranks."""
        if self.world == 1 and not (force and self.dist.is_initialized()):
            return self.local.search_device(q, k)
        nq = int(q.shape[0])
        if not hasattr(self.local, "search_device_packed"):  # an injected test double (CPU tests): the generic exchange + merge_fn
            D, I = self.local.search(q, k)
            return self.search_gathered(D, I, force=force)
        block, _, _ = self.local.search_device_packed(q, k)
        return self.exchange_packed(block, nq, k)

    def exchange_packed(self, block, nq, k):
        """The exchange step alone: this rank's packed (D, I) block (IndexFlatIP.search_device_packed) -> merged lists, identical on all ranks."""
        gathered = all_gather_dim0(block, self.world, self.group)
        return topk_merge_packed(gathered, self.world, nq, k)

    def search_gathered(self, D, I, force=False):
        """Exchange this rank's (D, I) [nq, k] with every other rank and merge; identical on all ranks.
        force: run the collective and the merge even in a one-rank group (self-test of the RCCL path on a 1-GPU box)."""
        if self.world == 1 and not (force and self.dist.is_initialized()):
            return D, I
        as_numpy = isinstance(D, np.ndarray)
        Dt = torch.from_numpy(D) if as_numpy else D
        It = torch.from_numpy(I) if as_numpy else I
        nq, k = Dt.shape
        # one packed buffer -> ONE collective per hop: scores as int32 bit patterns next to the ids
        packed = torch.stack([Dt.contiguous().view(torch.int32).to(torch.int64), It.contiguous()], 0)
        # concatenated along dim 0 (the layout both gloo and RCCL accept), viewed as [world, 2, nq, k]
        gathered = all_gather_dim0(packed, self.world, self.group).view(self.world, 2, nq, k)
        Dp = gathered[:, 0].to(torch.int32).view(torch.float32).reshape(self.world, nq, k)
        Ip = gathered[:, 1].reshape(self.world, nq, k)
        Dm, Im = self.merge_fn(Dp.contiguous(), Ip.contiguous())
        if as_numpy:
            return Dm.cpu().numpy(), Im.cpu().numpy()
        return Dm, Im

"""Drop-in for /root/reference/scripts/encode_corpus.py: encode a JSONL/TSV corpus into `<embed_save_path>.npy`
([N, hidden] float32) and `<embed_save_path>/id2doc.json`, same flags (config.encode_args).

    python scripts/encode_corpus.py --do_predict --predict_batch_size 1000 --model_name roberta-base \
        --predict_file ${CORPUS_PATH} --init_checkpoint ${MODEL_CHECKPOINT} --embed_save_path ${SAVE_PATH} \
        --fp16 --max_c_len 300 --num_workers 20

Instead of DataParallel's per-step parameter broadcast (reference :85-89), weights are replicated once per GPU
and, under torch.distributed.run, the corpus is split by contiguous row ranges across ranks; each rank writes
its rows into one shared memory-mapped .npy, so the full matrix never has to sit in host RAM (reference :93,110).
`--fp16` is accepted: fp16-operand / fp32-accumulate numerics (apex O1) are what the HIP encoder always runs.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from .config import encode_args
from .data import EmDataset, em_collate
from .retriever import RobertaConfig, RobertaCtxEncoder, load_saved, move_to_cuda


class _Indexed(torch.utils.data.Dataset):
    """Items of `dataset[lo:hi]` together with their global row index."""

    def __init__(self, dataset, lo, hi):
        self.dataset, self.lo, self.hi = dataset, lo, hi

    def __len__(self):
        return self.hi - self.lo

    def __getitem__(self, i):
        return self.lo + i, self.dataset[self.lo + i]


class LengthBucketCollate:
    """Collate one WINDOW of tokenised passages into length-homogeneous batches: sort the window by token count, cut it into
    batches of `batch_size`, em_collate each. Returns [(row indices int64, batch dict)]. Every passage keeps its own output
    row, so the saved matrix is the reference's; what changes is that a batch's padded width (and with it the encoder's
    hipGraph shape bucket) follows the length distribution instead of the longest outlier of a random batch."""

    def __init__(self, batch_size):
        self.batch_size = batch_size

    def __call__(self, samples):
        order = sorted(range(len(samples)), key=lambda j: (int(samples[j][1]["input_ids"].numel()), samples[j][0]))
        # COMPACT batches (round 3): a worker hands the main process the tokens of a batch as ONE int32 vector + the lengths instead of two
        # padded int64 matrices (ids, mask) -- 7-10 x fewer bytes through the DataLoader's shared memory, where the main process pays a page
        # fault per 4 KiB it touches (measured: 14-39 ms per batch of ~5 MB, the whole gap between 14 k passages/s end to end and 30 k on the
        # device). expand_compact() rebuilds exactly em_collate's (input_ids, input_mask) on the device. Only for the plain case: every
        # attention mask all ones, no token_type_ids; anything else takes em_collate as before.
        plain = all("token_type_ids" not in sm[1] and bool(sm[1]["attention_mask"].all()) for sm in samples)
        out = []
        for s in range(0, len(order), self.batch_size):
            sel = order[s:s + self.batch_size]
            rows = torch.tensor([samples[j][0] for j in sel], dtype=torch.int64)
            if plain:
                toks = [samples[j][1]["input_ids"].reshape(-1) for j in sel]
                out.append((rows, {"_compact_flat": torch.cat(toks).to(torch.int32), "_compact_lens": torch.tensor([t.numel() for t in toks], dtype=torch.int32)}))
            else:
                out.append((rows, em_collate([samples[j][1] for j in sel])))
        return out


def expand_compact(batch, to_device):
    """A batch of LengthBucketCollate -> the dict em_collate would have built ((B, L) int64 `input_ids` padded with 0, `input_mask`), made on
    the device `to_device` moves tensors to; non-compact batches just go through `to_device`."""
    if "_compact_flat" not in batch:
        return to_device(batch)
    lens_h = batch["_compact_lens"]
    B, L = int(lens_h.numel()), int(lens_h.max()) if lens_h.numel() else 0
    moved = to_device({"flat": batch["_compact_flat"], "lens": lens_h})
    flat, lens = moved["flat"], moved["lens"]
    mask = torch.arange(L, device=flat.device, dtype=torch.int32)[None, :] < lens[:, None]
    ids = torch.zeros((B, L), dtype=torch.int64, device=flat.device)
    ids.masked_scatter_(mask, flat.to(torch.int64))  # row-major order = the order the rows were concatenated in
    return {"input_ids": ids, "input_mask": mask.to(torch.int64)}


def shard_range(n, world, rank):
    """Contiguous row block of `rank` (the same rule as index.shard_bounds): [lo, hi)."""
    per = -(-n // world)
    return min(n, rank * per), min(n, (rank + 1) * per)


class DeviceStager:
    """Host -> device of a batch dict through TWO reusable pinned staging buffers and reusable device buffers (round 3).
    `x.cuda()` per tensor (reference utils.py:24-41, `move_to_cuda`) allocates a device tensor and, with a pinning DataLoader, a
    pinned host block per tensor and batch; batch shapes differ from batch to batch (every batch is padded to its own longest
    passage), so the caching allocators keep missing and the calls fall through to hipMalloc / hipHostMalloc / hipHostFree --
    measured on the box: 14-39 ms of host time per batch of ~5 MB with the GPU idle, i.e. the corpus encoder ran at 14 k passages/s
    end to end against 30 k on the device (scripts/measure/gpu_encode_corpus_probe.py). Here nothing is allocated in the steady state:
    the tensors of a batch are packed into one pinned buffer (a host memcpy), go over in ONE asynchronous copy into a device buffer
    of the same layout, and the views handed to the model alias that buffer. Stream order protects the device buffer (the copy of
    batch i+1 is queued behind the forward of batch i); an event per pinned buffer protects the host side."""

    def __init__(self, device):
        self.device = device
        self.host = [None, None]
        self.dev = [None, None]
        self.ev = [None, None]
        self.i = 0

    def __call__(self, batch):
        if len(batch) == 0:
            return {}
        items = [(k, v) for k, v in batch.items() if torch.is_tensor(v)]
        rest = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
        offs, total = [], 0
        for _, v in items:
            offs.append(total)
            total += -(-v.numel() * v.element_size() // 256) * 256
        j = self.i
        self.i ^= 1
        if self.host[j] is None or self.host[j].numel() < total:
            cap = max(total, 1 << 20) * 5 // 4
            self.host[j] = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
            self.dev[j] = torch.empty(cap, dtype=torch.uint8, device=self.device)
            self.ev[j] = None
        if self.ev[j] is not None:
            self.ev[j].synchronize()  # the copy that last read this pinned buffer (two batches ago) has finished
        host, dev = self.host[j], self.dev[j]
        out = dict(rest)
        for (k, v), o in zip(items, offs):
            nb = v.numel() * v.element_size()
            # numpy's memcpy: torch's CPU copy_ forks its intra-op thread pool (256 threads on the MI355X hosts) for a copy of this size and waits for
            # cores the DataLoader workers occupy -- 3-4 ms per 280 KB measured in the eval CLI (pipeline.py)
            np.copyto(host[o:o + nb].view(v.dtype).view(v.shape).numpy(), v.detach().contiguous().numpy())
            out[k] = dev[o:o + nb].view(v.dtype).view(v.shape)
        dev[:total].copy_(host[:total], non_blocking=True)
        self.ev[j] = torch.cuda.Event()
        self.ev[j].record()
        return out


def predict(model, eval_dataloader, out, out_bf16=None, to_device=move_to_cuda):
    """reference :95-113 -- batches -> model(batch)['embed'] -> rows of `out` (streamed into the memory-mapped matrix, not
    torch.cat'ed in host RAM). The loader yields windows of (row indices, batch) pairs (LengthBucketCollate).
    out_bf16 (optional): a uint16 matrix receiving the same rows rounded to bf16 (round-to-nearest-even bit patterns).
    The device -> host copy and the host-side write of batch i overlap the forward of batch i+1 (pinned staging buffer + event);
    the reference's `.cpu()` per batch serialises them."""
    model.eval()
    n = 0
    pending = None
    if to_device is move_to_cuda and torch.cuda.is_available():
        to_device = DeviceStager(torch.device("cuda", torch.cuda.current_device()))
    stage = {}  # reusable pinned D2H buffers, two per (shape, dtype): batch i's is flushed before batch i+2 needs it

    def pinned(shape, dtype, slot):
        key = (tuple(shape), dtype, slot)
        if key not in stage:
            if len(stage) > 16:
                stage.clear()
            stage[key] = torch.empty(shape, dtype=dtype, pin_memory=True)
        return stage[key]

    def flush(p):
        rows, host, host16, ev, _ = p
        if ev is not None:
            ev.synchronize()
        out[rows.numpy()] = host.numpy()
        if host16 is not None:
            out_bf16[rows.numpy()] = host16.numpy().view(np.uint16)

    for window in eval_dataloader:
        for rows, batch in window:
            batch_to_feed = expand_compact(batch, to_device)
            with torch.no_grad():
                e = model(batch_to_feed)["embed"]
            if e.is_cuda:
                slot = 0 if pending is None else 1 - pending[4]
                host = pinned(e.shape, e.dtype, slot)
                host.copy_(e, non_blocking=True)
                host16 = None
                if out_bf16 is not None:
                    host16 = pinned(e.shape, torch.int16, slot)
                    host16.copy_(e.to(torch.bfloat16).view(torch.int16), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:  # CPU stand-in encoders of the tests
                host, ev, slot = e, None, 0
                host16 = e.to(torch.bfloat16).view(torch.int16) if out_bf16 is not None else None
            if pending is not None:
                flush(pending)
            pending = (rows, host, host16, ev, slot)
            n += int(e.shape[0])
    if pending is not None:
        flush(pending)
    return n


def encode_shard(model, dataset, args, rank, world, hidden, barrier=None, to_device=move_to_cuda):
    """This rank's share of the job (what replaces the reference's DataParallel wrap, :85-89): rows [lo, hi) of the corpus
    through `model`, written into ONE memory-mapped `<embed_save_path>.npy` shared by all ranks (rank 0 creates it; the
    barrier orders creation before the first write and the last write before anyone reads). Returns (path, rows written)."""
    n = len(dataset)
    lo, hi = shard_range(n, world, rank)
    window = max(1, int(getattr(args, "length_bucket_window", 1))) * args.predict_batch_size
    loader = DataLoader(_Indexed(dataset, lo, hi), batch_size=window, collate_fn=LengthBucketCollate(args.predict_batch_size),
                        pin_memory=False, num_workers=args.num_workers)  # predict() stages through its own two pinned buffers (DeviceStager)
    path = args.embed_save_path + ".npy"  # np.save appends .npy to the same string that names the id2doc directory (:93)
    side = args.embed_save_path + ".bf16.npy"
    if rank == 0:
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(n, hidden))
        del mm
        if args.save_bf16:
            mm = np.lib.format.open_memmap(side, mode="w+", dtype=np.uint16, shape=(n, hidden))
            del mm
    if barrier is not None:
        barrier()
    out = np.load(path, mmap_mode="r+")
    out16 = np.load(side, mmap_mode="r+") if args.save_bf16 else None
    done = predict(model, loader, out, out_bf16=out16, to_device=to_device)
    out.flush()
    if out16 is not None:
        out16.flush()
    if barrier is not None:
        barrier()
    return path, done


def main(argv=None, tokenizer=None):
    args = encode_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("encode_corpus needs a HIP device: there is no CPU path")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else max(args.local_rank, 0))
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl")
    if not args.predict_file:
        raise ValueError("If `do_predict` is True, then `predict_file` must be specified.")
    if "roberta" not in args.model_name and not os.path.isdir(args.model_name):
        raise SystemExit("only the RoBERTa encoder (RobertaCtxEncoder) is implemented; pass --model_name roberta-base or a local dir")
    cfg = RobertaConfig()
    if os.path.isdir(args.model_name):
        try:
            from transformers import AutoConfig
            cfg = AutoConfig.from_pretrained(args.model_name)
        except Exception:
            pass
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.model_name)
    model = RobertaCtxEncoder(cfg, args)
    dataset = EmDataset(tokenizer, args.predict_file, args.max_q_len, args.max_c_len, args.is_query_embed, args.embed_save_path,
                        write_id2doc=(rank == 0))
    assert args.init_checkpoint != ""
    model = load_saved(model, args.init_checkpoint, exact=False)
    model.to(torch.device("cuda"))

    barrier = torch.distributed.barrier if world > 1 else None
    path, _ = encode_shard(model, dataset, args, rank, world, cfg.hidden_size, barrier=barrier)
    n = len(dataset)
    if rank == 0:
        # what the rows are valid for: the encoder's numerics mode (mdr_encoder_config.residual_fp32) and the token caps; the eval CLI warns
        # when its query encoder runs another mode than the corpus was encoded under (within-noise ranking flips are then possible)
        import json
        with open(args.embed_save_path + ".meta.json", "w") as f:
            json.dump({"rows": n, "dim": int(cfg.hidden_size), "encoder_numerics": {"residual_fp32": int(model.residual_fp32)},
                       "max_c_len": int(args.max_c_len), "init_checkpoint": os.path.basename(args.init_checkpoint)}, f)
        print(torch.Size((n, cfg.hidden_size)))
    return path


if __name__ == "__main__":
    main()

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
        out = []
        for s in range(0, len(order), self.batch_size):
            sel = order[s:s + self.batch_size]
            out.append((torch.tensor([samples[j][0] for j in sel], dtype=torch.int64), em_collate([samples[j][1] for j in sel])))
        return out


def shard_range(n, world, rank):
    """Contiguous row block of `rank` (the same rule as index.shard_bounds): [lo, hi)."""
    per = -(-n // world)
    return min(n, rank * per), min(n, (rank + 1) * per)


def predict(model, eval_dataloader, out, out_bf16=None, to_device=move_to_cuda):
    """reference :95-113 -- batches -> model(batch)['embed'] -> rows of `out` (streamed into the memory-mapped matrix, not
    torch.cat'ed in host RAM). The loader yields windows of (row indices, batch) pairs (LengthBucketCollate).
    out_bf16 (optional): a uint16 matrix receiving the same rows rounded to bf16 (round-to-nearest-even bit patterns).
    The device -> host copy and the host-side write of batch i overlap the forward of batch i+1 (pinned staging buffer + event);
    the reference's `.cpu()` per batch serialises them."""
    model.eval()
    n = 0
    pending = None

    def flush(p):
        rows, host, host16, ev = p
        if ev is not None:
            ev.synchronize()
        out[rows.numpy()] = host.numpy()
        if host16 is not None:
            out_bf16[rows.numpy()] = host16.numpy().view(np.uint16)

    for window in eval_dataloader:
        for rows, batch in window:
            batch_to_feed = to_device(batch)
            with torch.no_grad():
                e = model(batch_to_feed)["embed"]
            if e.is_cuda:
                host = torch.empty(e.shape, dtype=e.dtype, pin_memory=True)
                host.copy_(e, non_blocking=True)
                host16 = None
                if out_bf16 is not None:
                    host16 = torch.empty(e.shape, dtype=torch.int16, pin_memory=True)
                    host16.copy_(e.to(torch.bfloat16).view(torch.int16), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:  # CPU stand-in encoders of the tests
                host, ev = e, None
                host16 = e.to(torch.bfloat16).view(torch.int16) if out_bf16 is not None else None
            if pending is not None:
                flush(pending)
            pending = (rows, host, host16, ev)
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
                        pin_memory=torch.cuda.is_available(), num_workers=args.num_workers)
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
        print(torch.Size((n, cfg.hidden_size)))
    return path


if __name__ == "__main__":
    main()

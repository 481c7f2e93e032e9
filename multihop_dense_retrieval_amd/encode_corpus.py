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


def predict(model, eval_dataloader, out, row0=0, out_bf16=None):
    """reference :95-113 -- batches -> model(batch)['embed'] -> rows of `out` (streamed, not torch.cat'ed).
    out_bf16 (optional): a uint16 matrix receiving the same rows rounded to bf16 (round-to-nearest-even bit patterns)."""
    model.eval()
    at = row0
    for batch in eval_dataloader:
        batch_to_feed = move_to_cuda(batch)
        with torch.no_grad():
            e = model(batch_to_feed)["embed"]
            embed = e.cpu().numpy()
            if out_bf16 is not None:
                out_bf16[at:at + embed.shape[0]] = e.to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16)
        out[at:at + embed.shape[0]] = embed
        at += embed.shape[0]
    return at - row0


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

    n = len(dataset)
    per = -(-n // world)
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    loader = DataLoader(torch.utils.data.Subset(dataset, range(lo, hi)), batch_size=args.predict_batch_size, collate_fn=em_collate,
                        pin_memory=True, num_workers=args.num_workers)
    path = args.embed_save_path + ".npy"  # np.save appends .npy to the same string that names the id2doc directory (:93)
    side = args.embed_save_path + ".bf16.npy"
    if rank == 0:
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(n, cfg.hidden_size))
        del mm
        if args.save_bf16:
            mm = np.lib.format.open_memmap(side, mode="w+", dtype=np.uint16, shape=(n, cfg.hidden_size))
            del mm
    if world > 1:
        torch.distributed.barrier()
    out = np.load(path, mmap_mode="r+")
    out16 = np.load(side, mmap_mode="r+") if args.save_bf16 else None
    predict(model, loader, out, row0=lo, out_bf16=out16)
    out.flush()
    if out16 is not None:
        out16.flush()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(torch.Size((n, cfg.hidden_size)))
    return path


if __name__ == "__main__":
    main()

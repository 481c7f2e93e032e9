"""Drop-in for /root/reference/scripts/eval/eval_mhop_retrieval.py: same positional arguments, flags, log
lines and JSONL output; the encoder and the index run on MI355X through libmdrhip.so.

    python scripts/eval/eval_mhop_retrieval.py ${EVAL_DATA} ${CORPUS_VECTOR_PATH} ${CORPUS_DICT} ${MODEL_CHECKPOINT} \
        --batch-size 100 --beam-size 1 --topk 1 --shared-encoder --model-name roberta-base --gpu --save-path ${OUT}

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N ...`; the index is row-sharded
over the ranks (one RCCL all_gather of per-shard top-k per hop), rank 0 logs and writes the output.

Differences from the reference, all deliberate: `--gpu` is implied (there is no CPU
path) and no device id is hard-coded; `--hnsw` is rejected (approximate search is out of scope);
`--save-index` is rejected (it wrote a FAISS file to a hard-coded path); the tokenizer is loaded from
`--model-name` as a LOCAL directory when there is no network.
"""
import argparse
import json
import logging
import os

import numpy as np
import torch

from . import mhop
from .index import IndexFlatIP, ShardedIndexFlatIP
from .retriever import RobertaConfig, RobertaRetriever, load_saved, move_to_cuda

logger = logging.getLogger()


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("raw_data", type=str, default=None)
    p.add_argument("indexpath", type=str, default=None)
    p.add_argument("corpus_dict", type=str, default=None)
    p.add_argument("model_path", type=str, default=None)
    p.add_argument("--topk", type=int, default=2, help="topk paths")
    p.add_argument("--num-workers", type=int, default=10)
    p.add_argument("--max-q-len", type=int, default=70)
    p.add_argument("--max-c-len", type=int, default=300)
    p.add_argument("--max-q-sp-len", type=int, default=350)
    p.add_argument("--batch-size", type=int, default=100)
    p.add_argument("--beam-size", type=int, default=5)
    p.add_argument("--model-name", type=str, default="roberta-base")
    p.add_argument("--gpu", action="store_true")
    p.add_argument("--save-index", action="store_true")
    p.add_argument("--only-eval-ans", action="store_true")
    p.add_argument("--shared-encoder", action="store_true")
    p.add_argument("--save-path", type=str, default="")
    p.add_argument("--stop-drop", default=0, type=float)
    p.add_argument("--hnsw", action="store_true")
    # extension (not in the reference): build the hop-2 inputs on the device from a token arena of the corpus
    # (tokenised once, cached next to the corpus dict) instead of host dict lookups + tokenizer between the hops
    p.add_argument("--hop2-on-device", action="store_true")
    return p


def _setup_logging():
    logger.setLevel(logging.INFO)
    if logger.hasHandlers():
        logger.handlers.clear()
    logger.addHandler(logging.StreamHandler())


def _load_config(model_name):
    """HF config from a local directory when one is given, roberta-base constants otherwise."""
    if os.path.isdir(model_name):
        try:
            from transformers import AutoConfig
            return AutoConfig.from_pretrained(model_name)
        except Exception:
            pass
    return RobertaConfig()


def _tokenize(tokenizer, texts, pairs, max_length):
    """`batch_encode_plus(x, max_length=n, pad_to_max_length=True, return_tensors="pt")` of transformers 2.11
    (eval_mhop_retrieval.py:148,168) in the >=4 spelling: longest-first truncation, right-pad to max_length."""
    if pairs is None:
        return tokenizer(texts, max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")
    a, b = [p[0] for p in pairs], [p[1] for p in pairs]
    return tokenizer(a, b, max_length=max_length, padding="max_length", truncation="longest_first", return_tensors="pt")


def load_index(indexpath, d=768):
    """`xb = np.load(indexpath).astype('float32'); index = IndexFlatIP(d); index.add(xb)` (reference :94,121-122)
    without the two 16 GB host copies: the .npy is memory-mapped and uploaded in chunks, each rank taking
    only its own rows."""
    xb = np.load(indexpath, mmap_mode="r")
    n = xb.shape[0]
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        index = ShardedIndexFlatIP(d, n)
        index.local.reserve(index.hi - index.lo)
        step = 1 << 18
        for lo in range(index.lo, index.hi, step):
            index.add_local(np.ascontiguousarray(xb[lo:min(index.hi, lo + step)]))
        return index
    index = IndexFlatIP(d)
    index.reserve(n)
    step = 1 << 18
    for lo in range(0, n, step):
        index.add(np.ascontiguousarray(xb[lo:lo + step]))
    return index


def main(argv=None, tokenizer=None):
    args = build_parser().parse_args(argv)
    _setup_logging()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
    if rank != 0:
        logger.setLevel(logging.WARNING)
    if args.hnsw:
        raise SystemExit("--hnsw (approximate HNSW search) is not implemented: this build is the exact flat-IP path only")
    if args.save_index:
        raise SystemExit("--save-index wrote a FAISS file to a hard-coded path in the reference; not supported here")
    if args.only_eval_ans:
        raise SystemExit("--only-eval-ans (answer-string recall, CPU regex tokenizer) is not part of the retrieval hot path yet")

    logger.info("Loading data...")
    with open(args.raw_data) as f:
        ds_items = [json.loads(line) for line in f.readlines()]

    logger.info("Loading trained model...")
    bert_config = _load_config(args.model_name)
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.model_name)
    model = RobertaRetriever(bert_config, args)
    model = load_saved(model, args.model_path, exact=False)
    model.to(torch.device("cuda"))
    model.eval()

    logger.info("Building index...")
    index = load_index(args.indexpath, d=bert_config.hidden_size)

    logger.info("Loading corpus...")
    id2doc = mhop.load_corpus_dict(args.corpus_dict)
    logger.info(f"Corpus size {len(id2doc)}")

    arena = None
    if args.hop2_on_device:
        from .arena import TokenArena
        cache = args.corpus_dict + ".arena.npz"
        if os.path.exists(cache):
            arena = TokenArena.load(cache)
        else:
            logger.info("Tokenising the corpus once for device-side hop-2 assembly...")
            arena = TokenArena.from_corpus(id2doc, tokenizer, roberta="roberta" in args.model_name, max_tokens=args.max_q_sp_len)
            if rank == 0:
                arena.save(cache)
        arena = arena.to(torch.device("cuda"))

    logger.info("Encoding questions and searching")
    questions = [mhop.strip_question(it["question"]) for it in ds_items]
    metrics, retrieval_outputs = [], []
    roberta = "roberta" in args.model_name
    for b_start in range(0, len(questions), args.batch_size):
        with torch.no_grad():
            batch_q = questions[b_start:b_start + args.batch_size]
            batch_ann = ds_items[b_start:b_start + args.batch_size]
            enc = move_to_cuda(dict(_tokenize(tokenizer, batch_q, None, args.max_q_len)))
            q_embeds = model.encode_q(enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids", None))
            D, I = index.search(q_embeds, args.beam_size)
            if arena is not None:
                # questions re-encoded without the hop-1 length cap so the pair sees the same tokens the tokenizer would
                qfull = move_to_cuda(dict(_tokenize(tokenizer, batch_q, None, args.max_q_sp_len)))
                ids2, mask2 = arena.assemble_hop2(qfull["input_ids"], qfull["attention_mask"], I, D, args.max_q_sp_len)
                q_sp_embeds = model.encode_q(ids2, mask2, None)
                D, I = D.cpu().numpy(), I.cpu().numpy()
            else:
                D, I = D.cpu().numpy(), I.cpu().numpy()
                pairs = mhop.build_hop2_pairs(batch_q, D, I, id2doc, roberta=roberta)
                enc2 = move_to_cuda(dict(_tokenize(tokenizer, None, pairs, args.max_q_sp_len)))
                q_sp_embeds = model.encode_q(enc2["input_ids"], enc2["attention_mask"], enc2.get("token_type_ids", None))
            D_, I_ = index.search(q_sp_embeds, args.beam_size)
            D_, I_ = D_.cpu().numpy(), I_.cpu().numpy()

            chains = mhop.rank_paths(D, I, D_, I_, args.beam_size, args.topk)
            for ann, ch in zip(batch_ann, chains):
                m = mhop.question_metrics(ch, ann["sp"], id2doc)
                m.update(question=ann["question"], type=ann["type"])
                metrics.append(m)
                retrieval_outputs.append(mhop.output_record(ann, ch, id2doc))

    if args.save_path != "" and rank == 0:
        with open(args.save_path, "w") as out:
            for rec in retrieval_outputs:
                out.write(json.dumps(rec) + "\n")

    for line in mhop.summary_lines(metrics):
        logger.info(line)
    return metrics, retrieval_outputs


if __name__ == "__main__":
    main()

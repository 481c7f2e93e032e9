"""Drop-in for /root/reference/scripts/eval/eval_mhop_retrieval.py: same positional arguments, flags, log
lines and JSONL output; the encoder and the index run on MI355X through libmdrhip.so.

    python scripts/eval/eval_mhop_retrieval.py ${EVAL_DATA} ${CORPUS_VECTOR_PATH} ${CORPUS_DICT} ${MODEL_CHECKPOINT} \
        --batch-size 100 --beam-size 1 --topk 1 --shared-encoder --model-name roberta-base --gpu --save-path ${OUT}

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N ...`; the index is row-sharded
over the ranks (one RCCL all_gather of per-shard top-k per hop), rank 0 logs and writes the output.

Differences from the reference, all deliberate: `--gpu` is implied (there is no CPU
path) and no device id is hard-coded; `--hnsw` is rejected (approximate search is out of scope);
`--save-index` is rejected (it wrote a FAISS file to a hard-coded path); the tokenizer is loaded from
`--model-name` as a LOCAL directory when there is no network. Additions (off by default): `--hop2-on-device`
(token arena + device-side hop-2 assembly), `--index-storage bf16` (+ bf16 sidecar), `--corpus-store`
(memory-mapped corpus instead of the JSON dict). `--only-eval-ans` is the reference's answer-recall mode.
"""
import argparse
import json
import logging
import os

import numpy as np
import torch

from . import answer_recall, mhop
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
    # additions of this build (defaults reproduce the reference)
    p.add_argument("--index-storage", choices=["f32", "f32-compact", "bf16"], default="f32",
                   help="f32-compact: no int8 screening copy (FAISS's 4 B/element instead of 5; same results, slower beam-1 searches); "
                        "bf16: keep the index rows as bf16 in HBM (2 B/element); reads <index>.bf16.npy when it exists")
    p.add_argument("--corpus-store", action="store_true",
                   help="use (build on first use) the memory-mapped <corpus_dict>.store instead of parsing the JSON dict")
    # extension (not in the reference): build the hop-2 inputs on the device from a token arena of the corpus
    # (tokenised once, cached next to the corpus dict) instead of host dict lookups + tokenizer between the hops
    p.add_argument("--hop2-on-device", action="store_true")
    # extension: software-pipelined batch loop (hop 2 of batch i beside hop 1 of batch i+1; identical results)
    p.add_argument("--pipeline-batches", action="store_true")
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
    (eval_mhop_retrieval.py:148,168): `<s> q </s>` / `<s> q </s></s> d </s>`, longest-first truncation, right-pad to
    max_length. RoBERTa-family tokenizers (the reference's path): single texts go through the installed tokenizer's own call
    with 2.11's prefix space in front (data.prefix_space_2_11), pairs through data.encode_pairs_2_11, which also keeps the
    reference's (slow-tokenizer) truncation rule for odd token budgets. Any other family (the reference's `else` branches for
    BERT-style models): the installed tokenizer's own pair call, `[CLS] a [SEP] b [SEP]` with token_type_ids."""
    from .data import encode_pairs_2_11, is_roberta_family, prefix_space_2_11
    if not is_roberta_family(tokenizer):
        if pairs is None:
            return tokenizer(list(texts), max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")
        return tokenizer([p[0] for p in pairs], [p[1] for p in pairs], max_length=max_length, padding="max_length",
                         truncation="longest_first", return_tensors="pt")
    if pairs is None:
        return tokenizer([prefix_space_2_11(t) for t in texts], max_length=max_length, padding="max_length", truncation=True, return_tensors="pt")
    ids, mask = encode_pairs_2_11(tokenizer, [p[0] for p in pairs], [p[1] for p in pairs], max_length, True)
    return {"input_ids": torch.tensor(ids, dtype=torch.int64), "attention_mask": torch.tensor(mask, dtype=torch.int64)}


def load_corpus(corpus_dict, use_store, rank=0, world=1):
    """The corpus dict of the reference (:131-135), or its memory-mapped store (built once by rank 0)."""
    if not use_store:
        return mhop.load_corpus_dict(corpus_dict)
    from .corpus_store import build_store
    store = corpus_dict + ".store"
    if rank == 0 and (not os.path.exists(store) or os.path.getmtime(store) < os.path.getmtime(corpus_dict)):
        logger.info("Building the corpus store once...")
        build_store(corpus_dict, store)
    if world > 1:
        torch.distributed.barrier()
    return mhop.load_corpus_dict(store)


def bf16_sidecar_path(indexpath):
    """`<name>.bf16.npy` next to `<name>.npy`: the same matrix as uint16 bf16 bit patterns (encode_corpus --save_bf16)."""
    return (indexpath[:-4] if indexpath.endswith(".npy") else indexpath) + ".bf16.npy"


def load_index(indexpath, d=768, storage="f32"):
    """`xb = np.load(indexpath).astype('float32'); index = IndexFlatIP(d); index.add(xb)` (reference :94,121-122)
    without the two 16 GB host copies: the .npy is memory-mapped and uploaded in chunks, each rank taking
    only its own rows. storage="bf16": rows are kept as bf16 in HBM (2 B/element); when a bf16 sidecar of the matrix
    exists it is read instead of the fp32 file (half the disk and PCIe bytes, identical index contents)."""
    side = bf16_sidecar_path(indexpath)
    use_side = storage == "bf16" and os.path.exists(side)
    xb = np.load(side if use_side else indexpath, mmap_mode="r")
    n = xb.shape[0]

    def chunk(lo, hi):
        if use_side:  # copy out of the read-only map: torch wants a writable buffer
            return torch.from_numpy(np.array(xb[lo:hi]).view(np.int16)).view(torch.bfloat16)
        return np.ascontiguousarray(xb[lo:hi])

    kw = {"storage": "bf16"} if storage == "bf16" else {"storage": "compact"} if storage == "f32-compact" else {}
    step = 1 << 18
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        index = ShardedIndexFlatIP(d, n, local_index=IndexFlatIP(d, **kw))
        index.local.reserve(index.hi - index.lo)
        for lo in range(index.lo, index.hi, step):
            index.add_local(chunk(lo, min(index.hi, lo + step)))
        return index
    index = IndexFlatIP(d, **kw)
    index.reserve(n)
    for lo in range(0, n, step):
        index.add(chunk(lo, min(n, lo + step)))
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

    logger.info("Loading data...")
    with open(args.raw_data) as f:
        ds_items = [json.loads(line) for line in f.readlines()]
    if args.only_eval_ans:  # eval_mhop_retrieval.py:76-77: yes/no questions cannot be matched against passage text
        ds_items = [it for it in ds_items if it["answer"][0] not in ["yes", "no"]]

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
    index = load_index(args.indexpath, d=bert_config.hidden_size, storage=args.index_storage)

    logger.info("Loading corpus...")
    id2doc = load_corpus(args.corpus_dict, args.corpus_store, rank, world)
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

    def hop1_inputs(batch_q):
        return move_to_cuda(dict(_tokenize(tokenizer, batch_q, None, args.max_q_len)))

    def hop2_embeds(batch_q, D, I):
        """Hop-2 query embeddings of one batch from its hop-1 results (device tensors). Returns (q_sp_embeds, D, I as numpy);
        D carries the -inf of empty passages afterwards, as in the reference (:162-165)."""
        if arena is not None:
            # questions re-encoded without the hop-1 length cap so the pair sees the same tokens the tokenizer would
            qfull = move_to_cuda(dict(_tokenize(tokenizer, batch_q, None, args.max_q_sp_len)))
            ids2, mask2 = arena.assemble_hop2(qfull["input_ids"], qfull["attention_mask"], I, D, args.max_q_sp_len)
            return model.encode_q(ids2, mask2, None), D.cpu().numpy(), I.cpu().numpy()
        D, I = D.cpu().numpy(), I.cpu().numpy()
        pairs = mhop.build_hop2_pairs(batch_q, D, I, id2doc, roberta=roberta)
        enc2 = move_to_cuda(dict(_tokenize(tokenizer, None, pairs, args.max_q_sp_len)))
        return model.encode_q(enc2["input_ids"], enc2["attention_mask"], enc2.get("token_type_ids", None)), D, I

    def finish_batch(batch_ann, D, I, D_, I_):
        chains = mhop.rank_paths(D, I, D_, I_, args.beam_size, args.topk)
        for ann, ch in zip(batch_ann, chains):
            if args.only_eval_ans:  # answer-string recall over the retrieved chains; nothing is saved (:208-217)
                metrics.append(answer_recall.answer_metrics(ann, ch, id2doc))
                continue
            m = mhop.question_metrics(ch, ann["sp"], id2doc)
            m.update(question=ann["question"], type=ann["type"])
            metrics.append(m)
            retrieval_outputs.append(mhop.output_record(ann, ch, id2doc))

    starts = list(range(0, len(questions), args.batch_size))
    if not args.pipeline_batches:
        for b_start in starts:
            with torch.no_grad():
                batch_q = questions[b_start:b_start + args.batch_size]
                batch_ann = ds_items[b_start:b_start + args.batch_size]
                enc = hop1_inputs(batch_q)
                q_embeds = model.encode_q(enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids", None))
                D, I = index.search(q_embeds, args.beam_size)
                q_sp_embeds, D, I = hop2_embeds(batch_q, D, I)
                D_, I_ = index.search(q_sp_embeds, args.beam_size)
                finish_batch(batch_ann, D, I, D_.cpu().numpy(), I_.cpu().numpy())
    elif starts:
        # Software-pipelined loop (same results, batch for batch): batches are independent, so while batch i is in hop 2 the
        # questions of batch i+1 are encoded on a side stream (second encoder lane) and ONE search call serves the hop-2
        # queries of batch i together with the hop-1 queries of batch i+1 (more than 128 queries go 256 per corpus pass).
        side = torch.cuda.Stream()
        with torch.no_grad():
            bq = questions[starts[0]:starts[0] + args.batch_size]
            enc = hop1_inputs(bq)
            D, I = index.search(model.encode_q(enc["input_ids"], enc["attention_mask"], None), args.beam_size)
            for n, b_start in enumerate(starts):
                batch_q = questions[b_start:b_start + args.batch_size]
                batch_ann = ds_items[b_start:b_start + args.batch_size]
                nxt = questions[starts[n + 1]:starts[n + 1] + args.batch_size] if n + 1 < len(starts) else None
                q_next = None
                if nxt is not None:
                    enc_n = hop1_inputs(nxt)
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        q_next = model.encode_q(enc_n["input_ids"], enc_n["attention_mask"], None, lane=1)
                q_sp_embeds, Dn, In = hop2_embeds(batch_q, D, I)
                nsp = q_sp_embeds.shape[0]
                if q_next is not None:
                    torch.cuda.current_stream().wait_stream(side)
                    q_next.record_stream(torch.cuda.current_stream())
                    Dc, Ic = index.search(torch.cat([q_sp_embeds, q_next], 0), args.beam_size)
                else:
                    Dc, Ic = index.search(q_sp_embeds, args.beam_size)
                finish_batch(batch_ann, Dn, In, Dc[:nsp].cpu().numpy(), Ic[:nsp].cpu().numpy())
                if q_next is not None:
                    D, I = Dc[nsp:].contiguous(), Ic[nsp:].contiguous()

    if args.save_path != "" and rank == 0:
        with open(args.save_path, "w") as out:
            for rec in retrieval_outputs:
                out.write(json.dumps(rec) + "\n")

    for line in (answer_recall.answer_summary_lines(metrics) if args.only_eval_ans else mhop.summary_lines(metrics)):
        logger.info(line)
    return metrics, retrieval_outputs


if __name__ == "__main__":
    main()

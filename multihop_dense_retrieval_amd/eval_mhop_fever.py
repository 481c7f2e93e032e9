"""Drop-in for /root/reference/scripts/eval/eval_mhop_fever.py: the same two-hop loop on FEVER claims with separate
beam widths per hop (`--beam-size-1`, `--beam-size-2`) and chains saved as (title, text) pairs. Same positional
arguments and flags; the encoder and the index run on MI355X through libmdrhip.so.

Reference lines: arguments :45-62, corpus dict as lists [title, text, is_intro] and `title2doc` :79-81, claims :100,
hop-1 search :112, pairs with the empty-passage rule :115-122, hop-2 search + reshape :130-133, path ranking :136-156,
records :159-169.

Deliberate differences: `--gpu` is implied and no device id is hard-coded; the output goes to `--save-path` as given
(the reference prefixes a private absolute directory, :172); `--model-name` must be a RoBERTa geometry (the reference's
default `bert-base-uncased` would need BERT position / segment semantics this encoder does not implement); no apex.
"""
import argparse
import json
import logging
import os

import torch

from . import mhop
from .eval_mhop_retrieval import _load_config, _setup_logging, _tokenize, load_index
from .retriever import RobertaRetriever, load_saved, move_to_cuda

logger = logging.getLogger()


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("raw_data", type=str, default=None)
    p.add_argument("indexpath", type=str, default=None)
    p.add_argument("corpus_dict", type=str, default=None)
    p.add_argument("model_path", type=str, default=None)
    p.add_argument("--topk", type=int, default=2, help="topk paths")
    p.add_argument("--num-workers", type=int, default=10)
    p.add_argument("--max-q-len", type=int, default=45)
    p.add_argument("--max-c-len", type=int, default=350)
    p.add_argument("--max-q-sp-len", type=int, default=400)
    p.add_argument("--batch-size", type=int, default=100)
    p.add_argument("--beam-size-1", type=int, default=5)
    p.add_argument("--beam-size-2", type=int, default=5)
    p.add_argument("--model-name", type=str, default="bert-base-uncased")
    p.add_argument("--gpu", action="store_true")
    p.add_argument("--shared-encoder", action="store_true")
    p.add_argument("--save-path", type=str, default="")
    p.add_argument("--stop-drop", default=0, type=float)
    p.add_argument("--index-storage", choices=["f32", "bf16"], default="f32")  # addition of this build, see eval_mhop_retrieval
    return p


class fever_docs_view:
    """`id2doc[str(id)]` of the FEVER corpus dict ([title, text, is_intro] lists, :79) as the {"title", "text"} mapping mhop.build_hop2_pairs reads."""

    def __init__(self, id2doc):
        self.id2doc = id2doc

    def __getitem__(self, key):
        v = self.id2doc[key]
        return {"title": v[0], "text": v[1]}


def fever_record(item, chains, id2doc, title2doc):
    """{id, claim, candidate_chains: [[(title, text), (title, text)], ...]} with the text looked up BY TITLE (:161-169)."""
    out = []
    for h1, h2, _ in chains:
        t1, t2 = id2doc[str(h1)][0], id2doc[str(h2)][0]
        out.append([(t1, title2doc[t1]), (t2, title2doc[t2])])
    return {"id": item["id"], "claim": item["claim"], "candidate_chains": out}


def main(argv=None, tokenizer=None):
    args = build_parser().parse_args(argv)
    _setup_logging()
    if "roberta" not in args.model_name:
        raise SystemExit(f"--model-name {args.model_name}: this encoder implements the RoBERTa geometry only (the reference's FEVER "
                         "commands use roberta-base)")
    logger.info("Loading data...")
    with open(args.raw_data) as f:
        ds_items = [json.loads(line) for line in f.readlines()]

    bert_config = _load_config(args.model_name)
    logger.info("Building index...")
    index = load_index(args.indexpath, d=bert_config.hidden_size, storage=args.index_storage)

    logger.info("Loading corpus...")
    with open(args.corpus_dict) as f:
        id2doc = json.load(f)
    title2doc = {item[0]: item[1] for item in id2doc.values()}
    docs_view = fever_docs_view(id2doc)
    logger.info(f"Corpus size {len(id2doc)}")

    logger.info("Loading trained model...")
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.model_name)
    model = RobertaRetriever(bert_config, args)
    model = load_saved(model, args.model_path, exact=False)
    model.to(torch.device("cuda"))
    model.eval()

    logger.info("Encoding claims and searching")
    claims = [it["claim"] for it in ds_items]
    b1, b2 = args.beam_size_1, args.beam_size_2
    retrieval_outputs = []
    for b_start in range(0, len(claims), args.batch_size):
        with torch.no_grad():
            batch_q = claims[b_start:b_start + args.batch_size]
            batch_ann = ds_items[b_start:b_start + args.batch_size]
            enc = move_to_cuda(dict(_tokenize(tokenizer, batch_q, None, args.max_q_len)))
            q_embeds = model.encode_q(enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids", None))
            D, I = index.search(q_embeds, b1)
            D, I = D.cpu().numpy(), I.cpu().numpy()
            # (claim, passage) pairs with the empty-passage rule (:115-122) -- the same lines as the HotpotQA script's, on list-valued entries
            pairs = mhop.build_hop2_pairs(batch_q, D, I, docs_view, roberta=True)  # ("roberta" in model_name is always true here)
            enc2 = move_to_cuda(dict(_tokenize(tokenizer, None, pairs, args.max_q_sp_len)))
            q_sp_embeds = model.encode_q(enc2["input_ids"], enc2["attention_mask"], enc2.get("token_type_ids", None))
            D_, I_ = index.search(q_sp_embeds, b2)
            D_, I_ = D_.cpu().numpy(), I_.cpu().numpy()
            chains = mhop.rank_paths(D, I, D_, I_, b1, args.topk, beam2=b2)
            if args.save_path != "":
                for ann, ch in zip(batch_ann, chains):
                    retrieval_outputs.append(fever_record(ann, ch, id2doc, title2doc))

    if args.save_path != "":
        os.makedirs(os.path.dirname(os.path.abspath(args.save_path)), exist_ok=True)
        with open(args.save_path, "w") as out:
            for rec in retrieval_outputs:
                out.write(json.dumps(rec) + "\n")
    return retrieval_outputs


if __name__ == "__main__":
    main()

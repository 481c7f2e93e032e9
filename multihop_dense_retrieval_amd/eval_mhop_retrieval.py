"""Drop-in for /root/reference/scripts/eval/eval_mhop_retrieval.py: same positional arguments, flags, log
lines and JSONL output; the encoder and the index run on MI355X through libmdrhip.so.

    python scripts/eval/eval_mhop_retrieval.py ${EVAL_DATA} ${CORPUS_VECTOR_PATH} ${CORPUS_DICT} ${MODEL_CHECKPOINT} \
        --batch-size 100 --beam-size 1 --topk 1 --shared-encoder --model-name roberta-base --gpu --save-path ${OUT}

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N ...`; the index is row-sharded over the
ranks AND the question batches are partitioned over them (rank r owns batches r, r+N, ...: 1/N of the encoder
forwards per GPU; per hop one all_gather of the query embeddings and one of the per-shard top-k lists, pipeline.py);
rank 0 collects the records in input order, logs and writes the output.

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
from .retriever import RobertaConfig, RobertaRetriever, load_saved, move_to_cuda  # noqa: F401  (move_to_cuda: re-exported, utils.py:24-41)

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
    # extension: software-pipelined batch loop (hop 2 of batch i beside hop 1 of batch i+D as two concurrent forwards + ONE corpus pass; identical results)
    p.add_argument("--pipeline-batches", action="store_true", help="(default since round 4; kept so that older command lines still parse)")
    p.add_argument("--no-pipeline-batches", action="store_true",
                   help="one corpus pass per hop and batch, hop-1 and hop-2 forwards one after the other (the fused form gives the same bytes of output)")
    # extension: batches in flight in the host/device software pipeline (pipeline.py; default: 2 with --hop2-on-device, else 8)
    p.add_argument("--inflight", type=int, default=None)
    # multi-GPU launches (torch.distributed.run): "nccl" is RCCL; gloo (+ --share-gpu: every rank on cuda:0) exists for the tests
    p.add_argument("--dist-backend", default="nccl")
    p.add_argument("--share-gpu", action="store_true")
    return p


def _setup_logging():
    logger.setLevel(logging.INFO)
    if logger.hasHandlers():
        logger.handlers.clear()
    logger.addHandler(logging.StreamHandler())


def _load_config(model_name):
    """HF config from a local directory when one is given, roberta-base constants otherwise."""
    if os.path.isdir(model_name):
        # `AutoConfig.from_pretrained(args.model_name)` (reference :80) reads <dir>/config.json; only the geometry is used here, so the file is read
        # directly (importing transformers costs the CLI 0.8-2.5 s of start-up; RobertaConfig ignores the keys it does not know)
        try:
            with open(os.path.join(model_name, "config.json")) as f:
                return RobertaConfig(**json.load(f))
        except (OSError, ValueError, TypeError):
            pass
        try:
            from transformers import AutoConfig
            return AutoConfig.from_pretrained(model_name)
        except Exception:
            pass
    return RobertaConfig()


from .data import tokenize_2_11 as _tokenize  # noqa: E402  (the tests and the FEVER script import it from here)


def load_corpus(corpus_dict, use_store, rank=0, world=1):
    """The corpus dict of the reference (:131-135), or its memory-mapped store (built once by rank 0)."""
    if not use_store:
        return mhop.load_corpus_dict(corpus_dict)
    from .corpus_store import build_store
    store = corpus_dict + ".store"
    def fresh():
        return os.path.exists(store) and os.path.getmtime(store) >= os.path.getmtime(corpus_dict)

    if rank == 0 and not fresh():
        logger.info("Building the corpus store once...")
        with _failure_marker(store):
            build_store(corpus_dict, store)  # (written to a temporary name and renamed: a reader never sees a partial file)
    if world > 1:  # the other ranks wait for the file, not for a collective: this runs before the process touches the device
        wait_for_file(fresh, store, "the corpus store")
    return mhop.load_corpus_dict(store)


class _failure_marker:
    """Rank 0 builds a shared file while the other ranks poll for it (wait_for_file): if the build raises, `<file>.failed` tells them to stop waiting."""

    def __init__(self, target):
        self.marker = target + ".failed"

    def __enter__(self):
        if os.path.exists(self.marker):
            os.remove(self.marker)

    def __exit__(self, et, ev, tb):
        if et is not None:
            try:
                with open(self.marker, "w") as f:
                    f.write(f"{et.__name__}: {ev}\n")
            except OSError:
                pass
        return False


def wait_for_file(ready, target, what, timeout=None, poll=0.2, beat=30.0, ok_marker=None, ok_grace=60.0, why=None):
    """Ranks other than 0: poll until `ready()`; a heartbeat line every `beat` seconds, RuntimeError when rank 0 left `<target>.failed` or after
    `timeout` seconds (MDR_SHARED_FILE_TIMEOUT, default 6 h -- tokenising / indexing a 5 M-passage corpus takes minutes, never hours). A file, not a
    collective: a pending RCCL collective is aborted by the process-group watchdog after its own timeout (10 min by default).
    `ok_marker`: a file rank 0 touches when ITS copy of the target is valid and it is NOT building. If that marker is this job's and `ready()` still
    fails `ok_grace` seconds later, nobody is going to write the file this rank is waiting for (a cache that is not on a shared filesystem, a transient read
    error): RuntimeError with `why()` (the last load error) instead of a 6-hour wait."""
    import time
    timeout = float(os.environ.get("MDR_SHARED_FILE_TIMEOUT", 6 * 3600)) if timeout is None else timeout
    t0 = last = time.monotonic()
    try:  # a marker an EARLIER job left behind is not this job's failure: only one written since this process started counts
        import psutil
        born = psutil.Process().create_time() - 1.0
    except Exception:
        born = time.time() - 1.0
    ok_seen = None
    while not ready():
        now = time.monotonic()
        if os.path.exists(target + ".failed") and os.path.getmtime(target + ".failed") >= born:
            raise RuntimeError(f"rank 0 failed to build {what} ({target}): {open(target + '.failed').read().strip()}")
        if ok_marker and os.path.exists(ok_marker) and os.path.getmtime(ok_marker) >= born:
            ok_seen = now if ok_seen is None else ok_seen
            if now - ok_seen > ok_grace:
                raise RuntimeError(f"rank 0 holds a valid copy of {what} and is not rebuilding it, but this rank cannot load {target}"
                                   f"{': ' + str(why()) if why else ''} (is the cache on a filesystem every rank sees?)")
        if now - t0 > timeout:
            raise RuntimeError(f"gave up waiting for {what} ({target}) after {timeout:.0f} s: is rank 0 alive?")
        if now - last >= beat:
            logging.getLogger().warning(f"rank {os.environ.get('RANK', '?')}: still waiting for rank 0 to write {what} ({target}), {now - t0:.0f} s")
            last = now
        time.sleep(poll)


def bf16_sidecar_path(indexpath):
    """`<name>.bf16.npy` next to `<name>.npy`: the same matrix as uint16 bf16 bit patterns (encode_corpus --save_bf16)."""
    return (indexpath[:-4] if indexpath.endswith(".npy") else indexpath) + ".bf16.npy"


def index_meta_path(indexpath):
    """`<name>.meta.json` next to `<name>.npy` (written by encode_corpus: numerics mode of the encoder that produced the rows)."""
    return (indexpath[:-4] if indexpath.endswith(".npy") else indexpath) + ".meta.json"


def check_index_numerics(indexpath, model):
    """Warn when the corpus rows were encoded under another residual-stream numerics mode than the query encoder runs (the default moved from mode 0 to
    the apex-O1-faithful mode 2 in round 4: an index encoded by an older build is searched with slightly different query embeddings). No file, no check."""
    try:
        with open(index_meta_path(indexpath)) as f:
            mode = json.load(f)["encoder_numerics"]["residual_fp32"]
    except (OSError, KeyError, ValueError, TypeError):
        return None
    mine = int(getattr(model, "residual_fp32", -1))
    if int(mode) != mine:
        logger.warning(f"index {indexpath} was encoded with residual_fp32={mode}, the query encoder runs residual_fp32={mine}: re-encode the corpus or set "
                       f"MDR_RESIDUAL_FP32={mode} for bit-consistent embeddings")
    return int(mode)


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
    step = 1 << 18 if use_side else 1 << 22  # fp32 map: the library pipelines the upload itself (pinned double buffer, copy stream): few, large calls
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


LAST_RUN = {}  # timing / counters of the last main() call of this process (bench.py --mode cli and the tests read it)
_MARKS = []    # (stage name, perf_counter) of the current main() call: the start-up breakdown of LAST_RUN["startup_s"]


def _mark(name):
    import time
    _MARKS.append((name, time.perf_counter()))


def _startup_breakdown():
    """{stage: seconds} between consecutive marks, plus how old the process was when main() started (interpreter start + imports: torch, transformers)."""
    out = {b[0]: round(b[1] - a[1], 4) for a, b in zip(_MARKS, _MARKS[1:])}
    try:
        import time
        import psutil
        out["process_before_main"] = round(_MARKS[0][2] - psutil.Process().create_time(), 4) if len(_MARKS[0]) > 2 else None  # interpreter start + imports
    except Exception:
        pass
    return out


def main(argv=None, tokenizer=None):
    import time
    del _MARKS[:]
    _MARKS.append(("main", time.perf_counter(), time.time()))
    args = build_parser().parse_args(argv)
    _setup_logging()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.hnsw:
        raise SystemExit("--hnsw (approximate HNSW search) is not implemented: this build is the exact flat-IP path only")
    if args.save_index:
        raise SystemExit("--save-index wrote a FAISS file to a hard-coded path in the reference; not supported here")
    if rank != 0:
        logger.setLevel(logging.WARNING)
    # the tokenizer and its worker processes come FIRST: the workers are forked before this process touches the device
    if tokenizer is None:
        from .data import load_tokenizer
        tokenizer = load_tokenizer(args.model_name)  # AutoTokenizer.from_pretrained, or the same BPE without importing transformers (data._LightBPE)
    from .pipeline import TokenizerPool
    pool = TokenizerPool(tokenizer, args.num_workers)
    _mark("tokenizer_and_workers")
    try:
        return _run(args, tokenizer, pool, world, rank)
    finally:
        pool.close()


def _run(args, tokenizer, pool, world, rank):
    import time

    from .pipeline import FinishPool, TwoHopPipeline, gather_results
    dist = torch.distributed

    logger.info("Loading data...")
    with open(args.raw_data) as f:
        ds_items = [json.loads(line) for line in f.readlines()]
    if args.only_eval_ans:  # eval_mhop_retrieval.py:76-77: yes/no questions cannot be matched against passage text
        ds_items = [it for it in ds_items if it["answer"][0] not in ["yes", "no"]]

    # Start-up order. The reference goes model -> index -> corpus (:80-135). Here everything that lives on the HOST comes first -- checkpoint into
    # host memory, corpus -- then the worker processes that finish batches are forked (they inherit the corpus and must not inherit a live HIP
    # runtime: forked after it, every device allocation of the parent -- workspace, hipGraph instantiation -- took ~50x longer, 6.5 s of captures
    # measured), and only then the process touches the device: process group, weights, index. The log lines are the reference's; "Loading corpus..."
    # now comes before "Building index...".
    _mark("questions")
    logger.info("Loading trained model...")
    bert_config = _load_config(args.model_name)
    model = RobertaRetriever(bert_config, args)
    model = load_saved(model, args.model_path, exact=False, map_location="cpu")

    _mark("checkpoint_to_host")
    logger.info("Loading corpus...")
    id2doc = load_corpus(args.corpus_dict, args.corpus_store, rank, world)
    logger.info(f"Corpus size {len(id2doc)}")
    _mark("corpus")

    class _Memo(dict):
        """Per-batch view of id2doc: a passage is looked up (corpus store: decoded from the map) once, however many chains name it."""

        def __missing__(self, key):
            v = self[key] = corpus[key]
            return v

    corpus = id2doc

    def finish_batch(batch_ann, D, I, D_, I_):
        """Path ranking, metrics and output records of one batch (eval_mhop_retrieval.py:181-258); runs in a finishing process (or, with
        --num-workers 0, on the finisher thread)."""
        chains = mhop.rank_paths(D, I, D_, I_, args.beam_size, args.topk)
        ms, recs = [], []
        id2doc = _Memo()
        for ann, ch in zip(batch_ann, chains):
            if args.only_eval_ans:  # answer-string recall over the retrieved chains; nothing is saved (:208-217)
                ms.append(answer_recall.answer_metrics(ann, ch, id2doc))
                continue
            m = mhop.question_metrics(ch, ann["sp"], id2doc)
            m.update(question=ann["question"], type=ann["type"])
            ms.append(m)
            recs.append(mhop.output_record(ann, ch, id2doc))
        return ms, recs

    finish_pool = FinishPool(finish_batch, 0 if args.num_workers <= 0 else max(1, min(4, args.num_workers // 4)))
    _mark("finish_workers_fork")
    try:
        return _run_on_device(args, tokenizer, pool, finish_pool, world, rank, ds_items, bert_config, model, id2doc, finish_batch)
    finally:
        finish_pool.close()


def _run_on_device(args, tokenizer, pool, finish_pool, world, rank, ds_items, bert_config, model, id2doc, finish_batch):
    import time

    from .pipeline import TwoHopPipeline, gather_results
    dist = torch.distributed
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(args.dist_backend)
    model.to(torch.device("cuda"))
    model.eval()
    _mark("device_init_and_weights")
    # The loop repeats two shapes (B x max_q_len, B*beam x max_q_sp_len): their hipGraphs are captured here, as part of loading the model (typical fills:
    # questions ~20 of 70 tokens, pairs ~60 % of 350). Any other shape (the ragged last batch) runs eagerly once: capturing it would cost more than it saves.
    # (the larger shape first: a lane's captures hold pointers into its workspace and are dropped when it grows)
    model.precapture(args.batch_size * args.beam_size, args.max_q_sp_len, 0.6, lane=0)
    model.precapture(args.batch_size, args.max_q_len, 0.3, lane=0)
    if not args.no_pipeline_batches:
        model.precapture(args.batch_size, args.max_q_len, 0.3, lane=1)

    _mark("graph_captures")
    logger.info("Building index...")
    index = load_index(args.indexpath, d=bert_config.hidden_size, storage=args.index_storage)
    check_index_numerics(args.indexpath, model)
    torch.cuda.synchronize()
    _mark("index_upload")

    roberta = "roberta" in args.model_name
    arena = None
    if args.hop2_on_device:
        from .arena import TokenArena, arena_tag
        cache = args.corpus_dict + ".arena.npz"
        tag = arena_tag(tokenizer, roberta, args.max_q_sp_len)
        load_err = [None]

        def load_arena():
            try:
                return TokenArena.load(cache, expect_tag=tag) if os.path.exists(cache) else None  # None: written under another tokenisation rule
            except (OSError, ValueError, EOFError) as e:
                if repr(e) != load_err[0]:
                    logger.warning(f"token arena {cache} could not be loaded ({e!r}): treated as stale")
                load_err[0] = repr(e)
                return None

        arena = load_arena()
        # No collective around the build: rank 0 tokenises when ITS view of the cache is stale and replaces the file atomically; the
        # other ranks poll for a file with the right tag. (A rank that already holds a valid arena keeps it: same tag = same tokens.)
        if arena is None and rank == 0:
            logger.info("Tokenising the corpus once for device-side hop-2 assembly...")
            with _failure_marker(cache):
                arena = TokenArena.from_corpus(id2doc, tokenizer, roberta=roberta, max_tokens=args.max_q_sp_len)
                arena.save(cache, tag=tag)
        elif rank == 0 and world > 1:
            try:  # "my copy is valid, I am not building": lets a rank that cannot load the file give up after a minute instead of polling for hours
                with open(cache + ".ok", "w") as f:
                    f.write(tag if isinstance(tag, str) else repr(tag))
            except OSError:
                pass
        elif arena is None:
            box = {}

            def ready():
                box["a"] = load_arena()
                return box["a"] is not None
            wait_for_file(ready, cache, "the token arena", poll=1.0, ok_marker=cache + ".ok", why=lambda: load_err[0])
            arena = box["a"]
        arena = arena.to(torch.device("cuda"))

    _mark("token_arena")
    logger.info("Encoding questions and searching")
    questions = [mhop.strip_question(it["question"]) for it in ds_items]

    pipe = TwoHopPipeline(model, index, pool, id2doc, finish_batch, batch_size=args.batch_size, beam=args.beam_size, max_q_len=args.max_q_len,
                          max_q_sp_len=args.max_q_sp_len, roberta=roberta, arena=arena, device=torch.device("cuda", torch.cuda.current_device()),
                          rank=rank, world=world, depth=args.inflight, fuse=not args.no_pipeline_batches, finish_pool=finish_pool)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    try:
        mine = pipe.run(questions, ds_items)
    finally:
        pipe.close()
    fence()
    loop_s = time.perf_counter() - t0
    _mark("batch_loop")
    results = gather_results(mine, world)
    LAST_RUN.clear()
    LAST_RUN.update(startup_s=_startup_breakdown(), loop_seconds=loop_s, questions=len(questions), world=world, rank=rank, stats=dict(pipe.stats),
                    encoder_forward_calls=model.forward_calls, encoder_forward_rows=model.forward_rows,
                    graph_captures=model.graph_captures, graph_replays=model.graph_replays)
    if results is None:  # ranks other than 0: their part went to rank 0
        return [m for _, (ms, _) in sorted(mine, key=lambda t: t[0]) for m in ms], [r for _, (_, rs) in sorted(mine, key=lambda t: t[0]) for r in rs]
    metrics = [m for ms, _ in results for m in ms]
    retrieval_outputs = [r for _, rs in results for r in rs]

    if args.save_path != "" and rank == 0:
        with open(args.save_path, "w") as out:
            for rec in retrieval_outputs:
                out.write(json.dumps(rec) + "\n")

    for line in (answer_recall.answer_summary_lines(metrics) if args.only_eval_ans else mhop.summary_lines(metrics)):
        logger.info(line)
    return metrics, retrieval_outputs


if __name__ == "__main__":
    main()

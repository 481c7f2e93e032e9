"""GPU (-m gpu): the drop-in CLI under torch.distributed.run with the question batches PARTITIONED over the ranks (pipeline.py): world 2
and world 8, the ranks sharing the box's one GPU over gloo (every N > 1 code path of the CLI except RCCL's transport: real encoder, real
row-sharded search, packed exchange, mdr_topk_merge_packed, result gathering). The JSONL must be byte-identical to the one-rank run and
each rank must have run only its share of the encoder forwards."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import seeded

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def toy_assets(tmp_path_factory, tiny_roberta_tokenizer):
    """A 2-layer hidden-768 checkpoint, a 1 003-passage corpus encoded by the drop-in encode_corpus, 57 questions, and a model directory
    holding config + tokenizer (what --model-name points at when there is no network)."""
    from multihop_dense_retrieval_amd import encode_corpus
    import transformers
    tmp = tmp_path_factory.mktemp("cli_multirank")
    tok = tiny_roberta_tokenizer
    geom = dict(seeded.TINY, hidden=768, heads=12, ffn=512, vocab=max(seeded.TINY["vocab"], len(tok)))
    sd = seeded.make_state_dict(41, geom)
    cfg_dir = tmp / "toy-roberta"
    transformers.RobertaConfig(vocab_size=geom["vocab"], hidden_size=768, num_hidden_layers=geom["layers"], num_attention_heads=12, intermediate_size=512,
                               max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1).save_pretrained(cfg_dir)
    tok.save_pretrained(str(cfg_dir))
    ckpt = tmp / "enc.pt"
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    rng = np.random.default_rng(3)
    words = [f"w{i}" for i in range(400)]
    docs = [{"title": f"T{i}", "text": " ".join(rng.choice(words, rng.integers(5, 40)))} for i in range(1003)]
    docs[11]["text"] = " "
    corpus = tmp / "corpus.jsonl"
    corpus.write_text("\n".join(json.dumps(d) for d in docs))
    save = tmp / "emb"
    path = encode_corpus.main(["--do_predict", "--predict_batch_size", "100", "--model_name", str(cfg_dir), "--predict_file", str(corpus), "--init_checkpoint", str(ckpt),
                               "--embed_save_path", str(save), "--fp16", "--max_c_len", "30", "--num_workers", "0"], tokenizer=tok)
    qs = [{"_id": f"q{i}", "question": " ".join(rng.choice(words, 6)) + "?", "answer": ["a"], "sp": [f"T{i}", f"T{i + 1}"], "type": "bridge" if i % 2 else "comparison"}
          for i in range(57)]
    data = tmp / "qas.json"
    data.write_text("\n".join(json.dumps(q) for q in qs))
    return {"tmp": tmp, "data": str(data), "index": path, "corpus_dict": str(save / "id2doc.json"), "ckpt": str(ckpt), "cfg_dir": str(cfg_dir), "tok": tok}


def cli_args(a, out, extra):
    return [a["data"], a["index"], a["corpus_dict"], a["ckpt"], "--batch-size", "10", "--beam-size", "3", "--topk", "4", "--model-name", a["cfg_dir"], "--gpu",
            "--save-path", str(out), "--max-q-len", "12", "--max-q-sp-len", "40"] + extra


@pytest.mark.parametrize("extra", [[], ["--hop2-on-device", "--no-pipeline-batches"]], ids=["default", "device+unfused"])
def test_one_rank_pipeline_variants_agree(toy_assets, extra, monkeypatch):
    """Worker processes / in-flight depth / fusion do not change a byte of the output (one rank)."""
    monkeypatch.setenv("MDR_ALLOW_LATE_FORK", "1")  # the worker-process run below is the point; this pytest process has long touched the device
    from multihop_dense_retrieval_amd import eval_mhop_retrieval
    a = toy_assets
    base = a["tmp"] / "base.jsonl"
    if not base.exists():
        eval_mhop_retrieval.main(cli_args(a, base, ["--num-workers", "0", "--inflight", "1"]), tokenizer=a["tok"])
    out = a["tmp"] / ("v%d.jsonl" % len(extra))
    eval_mhop_retrieval.main(cli_args(a, out, ["--num-workers", "2"] + extra), tokenizer=a["tok"])
    assert out.read_text() == base.read_text()
    run = eval_mhop_retrieval.LAST_RUN
    assert run["questions"] == 57 and run["stats"]["batches"] == 6 and run["encoder_forward_calls"] == 12


@pytest.mark.parametrize("world,port,extra", [(2, 29591, []), (8, 29592, ["--hop2-on-device"]), (2, 29593, ["--hop2-on-device", "--no-pipeline-batches"])],
                         ids=["w2-default", "w8-device", "w2-device+unfused"])
def test_question_partitioned_cli_is_byte_identical_to_one_rank(toy_assets, world, port, extra):
    from multihop_dense_retrieval_amd import eval_mhop_retrieval
    a = toy_assets
    base = a["tmp"] / "base.jsonl"
    if not base.exists():
        eval_mhop_retrieval.main(cli_args(a, base, ["--num-workers", "0", "--inflight", "1"]), tokenizer=a["tok"])
    out = a["tmp"] / f"w{world}_{len(extra)}.jsonl"
    stats = a["tmp"] / f"stats_w{world}_{len(extra)}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "scripts", "gpu_cli_multirank.py"), str(stats)] + cli_args(a, out, ["--num-workers", "1", "--dist-backend", "gloo", "--share-gpu"] + extra),
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert out.read_text() == base.read_text()  # byte-identical JSONL, records in input order
    nb = 6  # 57 questions in batches of 10
    fwd = []
    for rank in range(world):
        run = json.load(open(f"{stats}.rank{rank}.json"))
        mine = len(range(rank, nb, world))
        assert run["stats"]["batches"] == mine and run["encoder_forward_calls"] == 2 * mine, (rank, run)  # 1/W of the encoder forwards: hop 1 + hop 2 of OWN batches only
        fwd.append(run["encoder_forward_rows"])
    assert sum(fwd) == 57 * (1 + 3)  # every question encoded once at hop 1, beam times at hop 2, over all ranks
    assert "Evaluating 57 samples..." in r.stderr and "\tAvg P-EM:" in r.stderr

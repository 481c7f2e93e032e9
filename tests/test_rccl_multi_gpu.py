"""GPU (-m gpu), needs >= 2 visible MI355X (skipped on the 1-GPU boxes; self-proving the moment a multi-GPU node runs the
suite): the row-sharded path of BASELINE.json's north star under RCCL -- one process per GPU, one all_gather of per-shard
top-k per hop, deterministic merge -- against the single-index result on the same synthetic corpus, and the two-rank
corpus encoder against the one-rank one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL over xGMI); the N>1 logic is covered on CPU by the world_size-2 gloo tests")


def _run(n, extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the plain command, no launcher: bench.py --gpus N starts its own N ranks (bench.self_launch); `port` only keeps the old signature
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rows", "2000000", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("beam", [1, 4])
def test_two_shards_return_the_single_index_ids(tmp_path, beam):
    _need_two()
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    a = _run(1, ["--no-encoder", "--beam", str(beam), "--topk", str(beam), "--dump-ids", one], 29561)
    b = _run(2, ["--no-encoder", "--beam", str(beam), "--topk", str(beam), "--scaling", "strong", "--dump-ids", two], 29562)
    assert a["self_check"]["full_size_exact"] and b["self_check"]["full_size_exact"]
    za, zb = np.load(one), np.load(two)
    for k in ("I", "I2"):
        assert np.array_equal(za[k], zb[k]), k  # identical ids: merge rule = (score desc, id asc), same as one index
    for k in ("D", "D2"):
        assert np.abs(za[k] - zb[k]).max() <= 1e-3


def test_weak_scaling_line_with_encoder_is_exact_and_carries_the_strong_number():
    _need_two()
    r = _run(2, [], 29563)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["global_batch"] == 200
    assert r["self_check"]["full_size_exact"]
    assert r["strong_scaling"]["value"] > 0 and r["roofline"]["frac"] <= 1.0 and r["roofline_encoder"]["frac"] <= 1.0


def test_two_rank_encode_corpus_equals_one_rank(tmp_path, tiny_roberta_tokenizer):
    _need_two()
    from oracle import seeded
    tok_dir = tmp_path / "tok"
    tiny_roberta_tokenizer.save_pretrained(str(tok_dir))
    import transformers
    geom = dict(seeded.TINY, hidden=768, heads=12, ffn=512, vocab=max(seeded.TINY["vocab"], len(tiny_roberta_tokenizer)))
    transformers.RobertaConfig(vocab_size=geom["vocab"], hidden_size=768, num_hidden_layers=geom["layers"], num_attention_heads=12,
                               intermediate_size=512, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                               pad_token_id=1).save_pretrained(str(tok_dir))
    ckpt = tmp_path / "enc.pt"
    torch.save({k: torch.from_numpy(v) for k, v in seeded.make_state_dict(31, geom).items()}, ckpt)
    rng = np.random.default_rng(1)
    words = "the quick brown fox retrieval encoder index beam passage question answer Paris London film band".split()
    corpus = tmp_path / "corpus.jsonl"
    corpus.write_text("".join(json.dumps({"title": f"T{i}", "text": " ".join(rng.choice(words, rng.integers(3, 50)))}) + "\n" for i in range(301)))
    outs = []
    for n, port in ((1, 29564), (2, 29565)):
        save = str(tmp_path / f"emb{n}")
        cmd = [sys.executable]
        if n > 1:
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
        cmd += [os.path.join(ROOT, "scripts", "encode_corpus.py"), "--do_predict", "--predict_batch_size", "32", "--model_name", str(tok_dir),
                "--predict_file", str(corpus), "--init_checkpoint", str(ckpt), "--embed_save_path", save, "--fp16", "--max_c_len", "64",
                "--num_workers", "0"]
        r = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        outs.append(np.load(save + ".npy"))
        assert len(json.load(open(os.path.join(save, "id2doc.json")))) == 301
    assert outs[0].shape == (301, 768) and np.abs(outs[0] - outs[1]).max() <= 1e-5

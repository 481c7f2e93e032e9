"""GPU (-m gpu), one device: `python bench.py --gpus N` WITHOUT a launcher must start its own N ranks (it used to run one
rank and print n_gpus 1). Rehearsed exactly as the driver types it, on one GPU shared by the ranks over gloo; plus the two refusals -- a launcher
whose WORLD_SIZE disagrees with --gpus, and fewer visible devices than ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env_extra=None, expect_rc=0):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    if expect_rc == 0:
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, lines  # ONE JSON line, from rank 0
        return json.loads(lines[0])
    assert r.returncode != 0, r.stdout[-2000:]
    return r.stderr


COMMON = ["--rows", "200000", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]


def test_gpus_2_without_a_launcher_runs_two_ranks(tmp_path):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    a = _bench(["--gpus", "1", "--no-encoder", "--dump-ids", one] + COMMON)
    b = _bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--no-encoder", "--scaling", "strong", "--dump-ids", two] + COMMON)
    assert a["n_gpus"] == 1 and a["config"]["collective_world_size"] == 1
    assert b["n_gpus"] == 2 and b["config"]["collective_world_size"] == 2 and b["config"]["collective_backend"] == "gloo"
    assert b["config"]["shards"] == 2
    assert a["self_check"]["full_size_exact"] and b["self_check"]["full_size_exact"]
    za, zb = np.load(one), np.load(two)
    for k in ("I", "I2"):
        assert np.array_equal(za[k], zb[k]), k  # two row shards + all-gather + merge return the one-index ids
    for k in ("D", "D2"):
        assert np.abs(za[k] - zb[k]).max() <= 1e-3


def test_the_drivers_literal_command_weak_scaling_with_encoder():
    r = _bench(["--gpus", "2", "--share-gpu", "--backend", "gloo"] + COMMON)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["global_batch"] == 200
    assert r["config"]["collective_world_size"] == 2
    assert r["self_check"]["full_size_exact"]
    assert r["strong_scaling"]["value"] > 0


def test_a_launcher_with_another_world_size_is_refused():
    err = _bench(["--gpus", "4"] + COMMON, env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_rc=1)
    assert "WORLD_SIZE=1" in err and "--gpus 4" in err


def test_more_ranks_than_devices_is_refused_without_share_gpu():
    n = torch.cuda.device_count() + 1
    err = _bench(["--gpus", str(n)] + COMMON, expect_rc=1)
    assert "HIP device(s) visible" in err


def test_structured_mode_reports_both_corpora_with_exact_ids():
    """`bench.py --mode structured` (round 6; the same sub-results ride in every default line): the k = 1 search on a clustered corpus and on the HIP encoder's own
    outputs, here at 300 k rows -- int8 tier deciding, query split on for the encoder rows only, ids equal to the brute force up to exact copies, and the beam > 1 case."""
    r = _bench(["--mode", "structured", "--rows", "300000"])
    s = r["structured"]
    assert set(s) == {"clustered", "encoder_geometry"}
    for name, v in s.items():
        for nq in ("nq100", "nq200"):
            x = v[nq]
            assert x["int8_tier_decided"] and not x["exact_fallback_ran"], (name, nq, x)
            assert x["top1_agreement_up_to_exact_ties"] == 1.0 and x["returned_score_vs_bruteforce_maxabs"] <= 5e-3, (name, nq, x)
        assert v["k4_nq400"]["lists_equal_up_to_matmul_noise"] and not v["k4_nq400"]["exact_fallback_ran"], (name, v["k4_nq400"])
    assert s["encoder_geometry"]["stats"]["mean_norm"] > 2 * s["encoder_geometry"]["stats"]["centred_row_norm_mean"]  # the geometry the query split is for
    assert s["clustered"]["stats"]["mean_norm"] < 0.2 * s["clustered"]["stats"]["centred_row_norm_mean"]

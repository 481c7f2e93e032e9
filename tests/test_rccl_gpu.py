"""GPU (-m gpu): the RCCL ("nccl") code path on ONE GPU. A one-rank process group runs the collectives of the N-GPU
bench (all_gather_into_tensor of fp32 embeddings and of the packed int64 (score bits, id) lists), the merge kernel and
an encoder graph replay between collectives -- scripts/gpu_rccl_selftest.py. N > 1 itself is covered by the
world-size-2 gloo tests (tests/test_sharded_gloo.py); no multi-GPU box was available to this round."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_group_runs_the_sharded_search_path():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "scripts", "gpu_rccl_selftest.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "rccl selftest ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])

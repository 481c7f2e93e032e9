"""GPU (-m gpu): world-8 (and world-2) run of the row-sharded index with the REAL HIP search and the REAL mdr_topk_merge, the ranks
sharing the box's one GPU over gloo (scripts/gpu_sharded_selftest.py) -- every N > 1 code path except RCCL's transport itself, which
needs more than one device (tests/test_rccl_multi_gpu.py proves that the moment a multi-GPU node runs the suite)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(8, 29581), (2, 29582)])
def test_world_n_sharded_search_with_the_real_kernels_equals_one_index(world, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SELFTEST_ROWS="400000" if world == 8 else "100003")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "scripts", "gpu_sharded_selftest.py")], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and f"sharded selftest world={world} ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("ids==one-index True") == 6  # 2 query counts x 3 beams

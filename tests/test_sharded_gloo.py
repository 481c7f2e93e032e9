"""CPU (-m "not gpu"), world_size 2 over gloo: the N>1 path of the row-sharded index -- partitioning, global id
arithmetic, the single all_gather per search and the merge rule -- with the CPU oracle standing in for the
per-shard HIP search (injected through ShardedIndexFlatIP(local_index=..., merge_fn=...)). The same class runs
on RCCL with the real kernels on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShard:
    """Test double for IndexFlatIP: same surface (add / search / id_offset / reserve), CPU oracle inside."""

    def __init__(self, d):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import OracleLib
        self.lib, self.d, self.rows, self.id_offset = OracleLib(), d, [], 0

    def reserve(self, n):
        pass

    def add(self, x):
        self.rows.append(np.asarray(x, np.float32))

    def search(self, q, k):
        xb = np.concatenate(self.rows) if self.rows else np.zeros((0, self.d), np.float32)
        D, I = self.lib.search(np.asarray(q, np.float32), xb, k)
        I = np.where(I >= 0, I + self.id_offset, -1)
        return torch.from_numpy(D), torch.from_numpy(I)


def merge_double(Dp, Ip):
    """Host statement of mdr_topk_merge's rule: score desc, id asc, padding (-1) last."""
    P, nq, k = Dp.shape
    D = torch.full((nq, k), -torch.finfo(torch.float32).max)
    I = torch.full((nq, k), -1, dtype=torch.int64)
    for q in range(nq):
        ent = sorted((-float(Dp[p, q, e]), int(Ip[p, q, e])) for p in range(P) for e in range(k) if int(Ip[p, q, e]) >= 0)
        for j, (s, i) in enumerate(ent[:k]):
            D[q, j], I[q, j] = -s, i
    return D, I


def _worker(rank, world, port, N, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from multihop_dense_retrieval_amd.index import ShardedIndexFlatIP, shard_bounds
    from oracle import seeded
    d = 64
    xb = seeded.normal(0, "shard.xb", (N, d))
    dup = min(3, N - 2)
    xb[N - 1] = xb[dup]  # exact duplicate across shards: the lower global id must come first
    q = seeded.normal(1, "shard.q", (9, d))
    q[0] = xb[dup]
    idx = ShardedIndexFlatIP(d, N, local_index=OracleShard(d), merge_fn=merge_double)
    assert (idx.lo, idx.hi) == shard_bounds(N, world, rank) and idx.local.id_offset == idx.lo
    idx.add_from_global(xb)
    D, I = idx.search(q, k)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), D=D.numpy(), I=I.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,k", [(2, 1001, 5), (2, 7, 4), (2, 2, 3), (8, 1001, 8), (8, 5, 4)])
def test_n_rank_sharded_search_equals_single_index(tmp_path, oracle, world, N, k):
    """world 8 = the north star's node (some shards are EMPTY at N = 5: their lists are all padding)."""
    from oracle import seeded
    port = _free_port()
    mp.spawn(_worker, args=(world, port, N, k, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    for r in range(1, world):
        r1 = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(r0["I"], r1["I"]) and np.array_equal(r0["D"], r1["D"])  # every rank holds the same merged lists
    d = 64
    xb = seeded.normal(0, "shard.xb", (N, d))
    dup = min(3, N - 2)
    xb[N - 1] = xb[dup]
    q = seeded.normal(1, "shard.q", (9, d))
    q[0] = xb[dup]
    D, I = oracle.search(q, xb, k)
    assert np.array_equal(r0["I"], I) and np.allclose(r0["D"], D, atol=1e-5)
    if N > 4:
        assert list(r0["I"][0, :2]) == [dup, N - 1]


def test_shard_bounds_cover_rows_exactly():
    from multihop_dense_retrieval_amd.index import shard_bounds
    for n in (0, 1, 7, 8, 9, 5_233_329):
        for w in (1, 2, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all(lo <= hi for lo, hi in spans)

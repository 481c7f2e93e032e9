"""TEST INFRASTRUCTURE (CPU checker / baseline), not product code: the same FAISS flat inner-product search as
oracle/flat_ip_oracle.c, with the block product done by the BLAS behind numpy (OpenBLAS sgemm) -- which is how FAISS itself
computes it (faiss 1.6.x utils/distances.cpp knn_inner_product_blas: blocked sgemm + heap update; faiss is not in
/root/reference: conda `faiss-gpu`, unpinned, setup.sh:9). Used by bench.py's cpu_baseline leg because it is the fastest
faithful CPU form on the GPU box's host cores, and pinned by tests/golden/mips_*.npz like the C restatement.

Tie rule: score descending, then id ascending (FAISS keeps the first-seen = lowest id) -- on the COMPUTED scores: a BLAS
sums exact duplicate rows in position-dependent order, so their fp32 scores may differ by an ulp and swap (true of FAISS
over MKL/OpenBLAS as well; oracle/flat_ip_oracle.c is the bit-reproducible checker). Fewer than k rows: -FLT_MAX / -1.
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def search(x, xb, k, block_rows=65536):
    x = np.ascontiguousarray(x, np.float32)
    nq = x.shape[0]
    n = xb.shape[0]
    D = np.full((nq, k), -FLT_MAX, np.float32)
    I = np.full((nq, k), -1, np.int64)
    if k == 1:
        for j0 in range(0, n, block_rows):
            S = x @ xb[j0:j0 + block_rows].T  # sgemm
            a = S.argmax(1)                   # first maximum = lowest id inside the block
            s = S[np.arange(nq), a]
            better = s > D[:, 0]              # strict: an equal score in a later block does not replace
            D[better, 0] = s[better]
            I[better, 0] = a[better] + j0
        return D, I
    for j0 in range(0, n, block_rows):
        S = x @ xb[j0:j0 + block_rows].T
        kb = min(k, S.shape[1])
        thr = np.partition(S, S.shape[1] - kb, axis=1)[:, S.shape[1] - kb]  # the block's k-th largest score per query
        for r in range(nq):
            idx = np.nonzero(S[r] >= thr[r])[0]  # ascending ids, ties of the k-th score included
            cs = np.concatenate([D[r], S[r, idx]])
            ci = np.concatenate([I[r], idx.astype(np.int64) + j0])
            ci_key = np.where(ci < 0, np.iinfo(np.int64).max, ci)
            order = np.lexsort((ci_key, -cs.astype(np.float64)))[:k]  # score descending, then id ascending
            D[r], I[r] = cs[order], ci[order]
    return D, I


def usable_cpus():
    """Cores this process may actually use: the smaller of its affinity mask and its cgroup CPU quota (the GPU boxes run
    the job in a container limited to a fraction of the host's hardware threads; oversubscribing the quota throttles)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n

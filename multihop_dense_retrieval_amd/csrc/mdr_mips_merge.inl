// csrc/mdr_mips_merge.inl -- result kernels: empty / top-1 finalisation, the k-way merge of per-workgroup lists, the cross-shard merge (mdr_topk_merge).
// Included by mdr_mips.hip inside namespace mdr::{anonymous}.
__global__ void fill_empty_kernel(float* D, long long* I, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { D[i] = -FLT_MAX; I[i] = -1; }
}

// k == 1: best[nq] -> D, I
__global__ void finalize_top1_kernel(const u64* __restrict__ best, int nq, float* __restrict__ D, long long* __restrict__ I, long long id_offset) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    u64 key = best[q];
    unsigned row = key_row(key);
    if (key == 0ull || row == 0xFFFFFFFFu) { D[q] = -FLT_MAX; I[q] = -1; }
    else { D[q] = key_score(key); I[q] = id_offset + (long long)row; }
}

// general k: merge G per-workgroup lists of one query group. One 256-thread block per query.
__global__ void __launch_bounds__(256)
merge_lists_kernel(const u64* __restrict__ cand, const int* __restrict__ cand_cnt, const u64* __restrict__ cand_kth, int G, int qcap,
                   int cap, int k, float* __restrict__ D, long long* __restrict__ I, long long id_offset, const int* __restrict__ run_if,
                   const float* __restrict__ qscale /* per query of this group, or null: scores are in the caller's scale already */) {
    __shared__ u64 keys[kMergeLds];
    if (run_if && *run_if == 0) return;
    __shared__ u64 sel[kKMax];
    __shared__ int red[4];
    __shared__ u64 s_u64[4];
    __shared__ int s_n;
    const int ql = blockIdx.x;
    const int tid = threadIdx.x;
    float* Dq = D + (size_t)ql * k;
    long long* Iq = I + (size_t)ql * k;

    // lower bound on the global k-th key: the largest per-list k-th key
    u64 t0 = 0ull;
    int total = 0;
    for (int w = tid; w < G; w += 256) {
        u64 v = cand_kth[(size_t)w * qcap + ql];
        t0 = v > t0 ? v : t0;
        total += cand_cnt[(size_t)w * qcap + ql];
    }
    for (int o = 32; o > 0; o >>= 1) { u64 v = __shfl_xor(t0, o); t0 = v > t0 ? v : t0; }
    if ((tid & 63) == 0) s_u64[tid >> 6] = t0;
    total = block_sum_256(total, red);
    t0 = s_u64[0];
    for (int i = 1; i < 4; ++i) t0 = s_u64[i] > t0 ? s_u64[i] : t0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    // survivors (keys >= t0) -> LDS when they fit
    int surv = 0;
    for (int w = 0; w < G; ++w) {
        int c = cand_cnt[(size_t)w * qcap + ql];
        const u64* lst = cand + ((size_t)w * qcap + ql) * cap;
        for (int i = tid; i < c; i += 256) {
            u64 v = lst[i];
            if (v >= t0) {
                int pos = atomicAdd(&s_n, 1);
                if (pos < kMergeLds) keys[pos] = v;
                ++surv;
            }
        }
    }
    __syncthreads();
    const int S = s_n;
    const bool in_lds = S <= kMergeLds;
    const int kk = total < k ? total : k;  // how many real results exist
    (void)surv;

    // k-th largest survivor by bisection on the 64-bit key
    u64 t = 0ull;
    if (kk > 0) {
        for (int bit = 63; bit >= 0; --bit) {
            u64 c = t | (1ull << bit);
            int n = 0;
            if (in_lds) {
                for (int i = tid; i < S; i += 256) n += keys[i] >= c;
            } else {
                for (int w = 0; w < G; ++w) {
                    int cc = cand_cnt[(size_t)w * qcap + ql];
                    const u64* lst = cand + ((size_t)w * qcap + ql) * cap;
                    for (int i = tid; i < cc; i += 256) n += lst[i] >= c;
                }
            }
            n = block_sum_256(n, red);
            if (n >= kk) t = c;
        }
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (kk > 0) {
        if (in_lds) {
            for (int i = tid; i < S; i += 256)
                if (keys[i] >= t) sel[atomicAdd(&s_n, 1)] = keys[i];
        } else {
            for (int w = 0; w < G; ++w) {
                int cc = cand_cnt[(size_t)w * qcap + ql];
                const u64* lst = cand + ((size_t)w * qcap + ql) * cap;
                for (int i = tid; i < cc; i += 256)
                    if (lst[i] >= t) sel[atomicAdd(&s_n, 1)] = lst[i];
            }
        }
    }
    __syncthreads();
    // exactly kk selected; order by rank counting (keys are unique)
    for (int i = tid; i < k; i += 256) {
        if (i < kk) {
            u64 me = sel[i];
            int rank = 0;
            for (int j = 0; j < kk; ++j) rank += sel[j] > me;
            Dq[rank] = key_score(me) * (qscale ? qscale[ql] : 1.f);
            Iq[rank] = id_offset + (long long)key_row(me);
        } else {
            Dq[i] = -FLT_MAX;
            Iq[i] = -1;
        }
    }
}

// cross-shard merge (mdr_topk_merge): entries compare by (score desc, id asc, position asc)
__global__ void __launch_bounds__(256)
merge_parts_kernel(const float* __restrict__ Dp, const long long* __restrict__ Ip, int nparts, int nq, int k, float* __restrict__ D,
                   long long* __restrict__ I) {
    const int q = blockIdx.x;
    const int T = nparts * k;
    for (int i = threadIdx.x; i < k; i += 256) { D[(size_t)q * k + i] = -FLT_MAX; I[(size_t)q * k + i] = -1; }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += 256) {
        int p = i / k, e = i - p * k;
        size_t at = ((size_t)p * nq + q) * k + e;
        float s = Dp[at];
        long long id = Ip[at];
        if (id < 0) continue;
        int rank = 0;
        for (int j = 0; j < T; ++j) {
            int pj = j / k, ej = j - pj * k;
            size_t aj = ((size_t)pj * nq + q) * k + ej;
            long long idj = Ip[aj];
            if (idj < 0) continue;
            float sj = Dp[aj];
            rank += (sj > s) || (sj == s && (idj < id || (idj == id && j < i)));
        }
        if (rank < k) { D[(size_t)q * k + rank] = s; I[(size_t)q * k + rank] = id; }
    }
}

// cross-shard merge of PACKED per-rank blocks (mdr_topk_merge_packed): part p = [scores f32 nq*k | pad to 8 B | ids i64 nq*k], `part_stride` bytes apart
// (what one all_gather of every rank's search output delivers). The nparts*k candidates of a query are staged in LDS once; the same order rule as
// merge_parts_kernel (score desc, id asc, position asc), no assumption that a part is sorted. kMergePackedLds entries fit; larger T take the global path.
constexpr int kMergePackedLds = 4096;
__global__ void __launch_bounds__(256)
merge_packed_kernel(const char* __restrict__ parts, long long part_stride, long long ids_off, int nparts, int nq, int k, float* __restrict__ D,
                    long long* __restrict__ I) {
    __shared__ float s_sc[kMergePackedLds];
    __shared__ long long s_id[kMergePackedLds];
    const int q = blockIdx.x;
    const int T = nparts * k;
    const bool in_lds = T <= kMergePackedLds;
    for (int i = threadIdx.x; i < k; i += 256) { D[(size_t)q * k + i] = -FLT_MAX; I[(size_t)q * k + i] = -1; }
    auto score_at = [&](int i) { int p = i / k, e = i - p * k; return ((const float*)(parts + p * part_stride))[(size_t)q * k + e]; };
    auto id_at = [&](int i) { int p = i / k, e = i - p * k; return ((const long long*)(parts + p * part_stride + ids_off))[(size_t)q * k + e]; };
    if (in_lds)
        for (int i = threadIdx.x; i < T; i += 256) { s_sc[i] = score_at(i); s_id[i] = id_at(i); }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += 256) {
        const float s = in_lds ? s_sc[i] : score_at(i);
        const long long id = in_lds ? s_id[i] : id_at(i);
        if (id < 0) continue;
        int rank = 0;
        if (in_lds) {
            for (int j = 0; j < T; ++j) {
                const long long idj = s_id[j];
                const float sj = s_sc[j];
                rank += (idj >= 0) && ((sj > s) || (sj == s && (idj < id || (idj == id && j < i))));
            }
        } else {
            for (int j = 0; j < T; ++j) {
                const long long idj = id_at(j);
                if (idj < 0) continue;
                const float sj = score_at(j);
                rank += (sj > s) || (sj == s && (idj < id || (idj == id && j < i)));
            }
        }
        if (rank < k) { D[(size_t)q * k + rank] = s; I[(size_t)q * k + rank] = id; }
    }
}

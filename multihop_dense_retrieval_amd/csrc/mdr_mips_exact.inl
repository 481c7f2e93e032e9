// csrc/mdr_mips_exact.inl (part 1) -- wave-level candidate lists and the exact 3-MFMA stream kernel (the fallback behind every screening tier).
// Included by mdr_mips.hip inside namespace mdr::{anonymous}.
// ---- wave-level candidate list maintenance -----------------------------------------------------------
__device__ inline u64 load_key_l2(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Keep the k largest of list[0..c) (unique 64-bit keys), compacted to the front. Whole wave calls it
// with identical arguments. Returns the k-th largest key (0 if c < k). E*64 >= c.
template <int E>
__device__ inline u64 wave_select_topk(u64* list, int c, int k, int lane, int* new_count) {
    // (before any load: a load left pending on an early return would make the compiler guard every later VMEM op
    //  of the caller's loop with vmcnt(0) and drain the corpus DMA each stage)
    if (c < k) { *new_count = c; return 0ull; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's appends have reached L2
    u64 key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int idx = e * 64 + lane;
        key[e] = idx < c ? load_key_l2(list + idx) : 0ull;
    }
    u64 t = 0ull;
    for (int bit = 63; bit >= 0; --bit) {
        u64 cand = t | (1ull << bit);
        int n = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) n += __popcll(__ballot(key[e] >= cand));
        if (n >= k) t = cand;
    }
    // t is now the k-th largest key: exactly k keys are >= t
    int base = 0;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int e = 0; e < E; ++e) {
        bool p = key[e] >= t;
        u64 m = __ballot(p);
        if (p) list[base + __popcll(m & lt)] = key[e];
        base += __popcll(m);
    }
    *new_count = base;
    return t;
}

// Per-wave bookkeeping after a row-block: prune every list of this wave that could overflow on the
// next row-block (16 appends per query at most). cnt[] lives in LDS, one int per query of the wave.
template <int E, int CAP>
__device__ inline void wave_prune_if_needed(u64* wave_lists /* [16][CAP] */, int* wave_cnt /* LDS [16] */, int k, int lane,
                                            float& tau, bool force, u64* kth_out /* [16] or null */) {
    int c = wave_cnt[lane & 15];
    bool need = force ? true : (c > CAP - 16);
    unsigned m = (unsigned)(__ballot(need) & 0xFFFFull);  // lanes 0..15 <-> the wave's 16 queries
    while (m) {
        int qi = __builtin_ctz(m);
        m &= m - 1;
        int cq = __shfl(c, qi);
        int nc;
        u64 t = wave_select_topk<E>(wave_lists + (size_t)qi * CAP, cq, k, lane, &nc);
        if ((lane & 15) == qi) {
            if (t) tau = key_score(t);
            if (lane == qi) {
                wave_cnt[qi] = nc;
                if (kth_out) kth_out[qi] = t;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void consider(float s, unsigned row, bool valid, float tau, u64* my_list, int* my_cnt) {
    if (valid && s >= tau) {
        int pos = atomicAdd(my_cnt, 1);  // LDS atomic
        my_list[pos] = make_key(s, row);
    }
}

// ---- the stream kernel -------------------------------------------------------------------------------
#define MDR_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MDR_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
// cache policy of the corpus stream (aux bits of global_load_lds: 0 = default, 2 = nt). The corpus is read once per
// search and is 30x the Infinity Cache, so the stream is non-temporal: measured 1.698 vs 1.733 ms per 5M-row search
// (interleaved A/B of two builds of these sources, gpurun_out r02a; scripts/measure/gpu_ab.sh rebuilds the comparison).
#ifndef MDR_MIPS_DMA_AUX
#define MDR_MIPS_DMA_AUX 2
#endif


// DMA one row-block (hi plane then lo plane, NKB KiB each) into an LDS slot: 2*NKB pieces over 8 waves
template <int NKB>
__device__ __forceinline__ void issue_row_block(const char* __restrict__ Xhi, const char* __restrict__ Xlo, int rb, char* slot, int wave, int lane) {
    constexpr int CPW = NKB / 4;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int piece = wave * CPW + c;  // wave-uniform
        const char* plane = piece < NKB ? Xhi : Xlo;
        const int kb = piece < NKB ? piece : piece - NKB;
        const char* g = plane + ((size_t)rb * NKB + kb) * kFragBytes + lane * 16;
        __builtin_amdgcn_global_load_lds(MDR_GPTR(g), MDR_LPTR(slot + piece * kFragBytes), 16, 0, MDR_MIPS_DMA_AUX);
    }
}

template <int NKB, int KMODE>  // KMODE 0: k == 1 (register argmax)   1: 2 <= k <= 128 (candidate lists)
__global__ void __launch_bounds__(512, 2)
mips_stream_kernel(const char* __restrict__ Xhi, const char* __restrict__ Xlo, long long n_rows, int n_rb, const char* __restrict__ Qhi,
                   const char* __restrict__ Qlo, int nq, u64* __restrict__ best, u64* __restrict__ cand, int* __restrict__ cand_cnt,
                   u64* __restrict__ cand_kth, int k, const int* __restrict__ run_if, const float* __restrict__ qscale) {
    if (run_if && *run_if == 0) return;  // speculative screen pass succeeded: nothing to do
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RB_BYTES = NKB * 2 * kFragBytes;
    constexpr int CPW = NKB / 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    const int n_my = (n_rb - b + G - 1) / G;
    int* lds_cnt = (int*)(lds + 3 * RB_BYTES);  // [128] (KMODE 1 only)

    if (KMODE == 1) {
        if (threadIdx.x < kStreamQ) lds_cnt[threadIdx.x] = 0;
    }

    // start the corpus stream before anything else
    if (n_my > 0) issue_row_block<NKB>(Xhi, Xlo, b, lds, wave, lane);
    if (n_my > 1) issue_row_block<NKB>(Xhi, Xlo, b + G, lds + RB_BYTES, wave, lane);

    // this wave's queries: B operand fragments for all of K, resident for the whole kernel
    const bool wave_active = wave * 16 < nq;
    half8 qh[NKB], ql[NKB];
    {
        const size_t qoff = (size_t)wave * NKB * kFragBytes + lane * 16;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            qh[kb] = *(const half8*)(Qhi + qoff + kb * kFragBytes);
            ql[kb] = *(const half8*)(Qlo + qoff + kb * kFragBytes);
        }
        // Make the compiler retire these loads HERE: if they were still pending (in its scoreboard) at
        // loop entry it would put an s_waitcnt vmcnt(0) in front of the first MFMA of every iteration
        // and drain the in-flight row-block DMA each time.
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            asm volatile("" : "+v"(qh[kb]));
            asm volatile("" : "+v"(ql[kb]));
        }
    }
    const int qlocal = wave * 16 + (lane & 15);
    const bool q_valid = qlocal < nq;
    const unsigned sub_row = 4u * (unsigned)(lane >> 4);

    float best_s = -FLT_MAX;
    unsigned best_row = 0xFFFFFFFFu;
    float tau = -INFINITY;
    u64* wave_lists = nullptr;
    u64* my_list = nullptr;
    if (KMODE == 1) {
        wave_lists = cand + ((size_t)b * kStreamQ + (size_t)wave * 16) * kStreamCap;
        my_list = wave_lists + (size_t)(lane & 15) * kStreamCap;
    }

    for (int it = 0; it < n_my; ++it) {
        // stage `it` has landed (ours), everyone is done reading the slot we are about to refill
        if (it + 1 < n_my)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 2 < n_my) issue_row_block<NKB>(Xhi, Xlo, b + (it + 2) * G, lds + ((it + 2) % 3) * RB_BYTES, wave, lane);

        if (wave_active) {
            const char* p = lds + (it % 3) * RB_BYTES + lane * 16;
            f32x4 aH = {0.f, 0.f, 0.f, 0.f}, aC1 = {0.f, 0.f, 0.f, 0.f}, aC2 = {0.f, 0.f, 0.f, 0.f};
            // LDS -> register prefetch PF k-blocks ahead of the MFMAs that consume them
            constexpr int PF = (KMODE == 0) ? 3 : 2;  // KMODE 1 needs the registers for list maintenance
            half8 xh[PF], xl[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                xh[i] = *(const half8*)(p + i * kFragBytes);
                xl[i] = *(const half8*)(p + (NKB + i) * kFragBytes);
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const half8 ch = xh[kb % PF], cl = xl[kb % PF];
                if (kb + PF < NKB) {
                    xh[kb % PF] = *(const half8*)(p + (kb + PF) * kFragBytes);
                    xl[kb % PF] = *(const half8*)(p + (NKB + kb + PF) * kFragBytes);
                }
                aH = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, qh[kb], aH, 0, 0, 0);
                aC1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, qh[kb], aC1, 0, 0, 0);
                aC2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, ql[kb], aC2, 0, 0, 0);
            }
            // pin the issue order: 2*PF reads up front, then per k-block {2 reads for kb+PF, 3 MFMAs of kb}
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb + PF < NKB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            // C layout: lane holds rows 4*(lane>>4)+r (corpus), column lane&15 (query)
            const unsigned row0 = (unsigned)(b + it * G) * 16u + sub_row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = aH[r] + (aC1[r] + aC2[r]) * kLoInv;
                unsigned row = row0 + r;
                bool ok = (long long)row < n_rows;
                if (KMODE == 0) {
                    if (ok && s > best_s) { best_s = s; best_row = row; }
                } else {
                    consider(s, row, ok && q_valid, tau, my_list, lds_cnt + qlocal);
                }
            }
            if (KMODE == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wave_prune_if_needed<kStreamCap / 64, kStreamCap>(wave_lists, lds_cnt + wave * 16, k, lane, tau, false, nullptr);
            }
        }
    }

    if (KMODE == 0) {
        // back to the caller's scale (queries were pre-scaled by a power of two: exact, order-preserving). KMODE 1 lists stay
        // in the scaled domain and merge_lists_kernel multiplies at the output.
        const float sc = q_valid ? qscale[qlocal] : 1.f;
        u64 key = make_key(best_s > -FLT_MAX ? best_s * sc : best_s, best_row);
        // lanes l, l^16, l^32, l^48 hold the same query
        u64 o = __shfl_xor(key, 16);
        key = o > key ? o : key;
        o = __shfl_xor(key, 32);
        key = o > key ? o : key;
        if (lane < 16 && q_valid && best) atomicMax(best + qlocal, key);
    } else if (wave_active) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_prune_if_needed<kStreamCap / 64, kStreamCap>(wave_lists, lds_cnt + wave * 16, k, lane, tau, true,
                                                          cand_kth + (size_t)b * kStreamQ + wave * 16);
        if (lane < 16) cand_cnt[(size_t)b * kStreamQ + qlocal] = lds_cnt[qlocal];
    }
}

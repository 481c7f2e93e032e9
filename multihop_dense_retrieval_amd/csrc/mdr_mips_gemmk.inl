// csrc/mdr_mips_gemmk.inl -- the screen-k main pass for groups of 256 queries as the GEMM it is (round 5).
// Included by mdr_mips.hip inside namespace mdr::{anonymous}, after mdr_mips_screen_fp16.inl (list protocol, keys, merge_screenk_kernel).
//
// mips_screenk32_kernel keeps 32 queries per wave in registers and lets EVERY wave read EVERY corpus stage from the LDS: 8 x 48 KiB of fragment
// reads per 32 rows against 8 x 48 MFMAs of 32x32x16 -- 256 B/clk of LDS traffic asked for where the CU delivers 128, so the matrix pipe idles half
// the time (6.25 M bf16 rows, 800 queries, k = 100: 11.3 ms = 0.68 PFLOP/s where the encoder's 256x256 GEMMs sustain 1.0-1.1 on the same chip).
// At 256 queries per pass the contraction is a [rows, 768] x [768, 256] GEMM and gets a GEMM's structure:
//   * tile = 256 corpus rows x 256 queries; eight waves as 2 (rows) x 4 (queries), 128 x 64 outputs each = 32 accumulator tiles of
//     v_mfma_f32_16x16x32; BOTH operands come through the LDS, so a K-step of 32 costs the CU 8 x (8 + 4) KiB of fragment reads for 8 x 32 MFMAs: 96 B/clk;
//   * K = 768 in 24 steps of ONE k-block. The corpus operand leaves HBM (latency: microseconds under load), the query operand the XCD's L2: FOUR 16 KiB
//     slots of row fragments and FOUR of query fragments = 128 KiB, each slot refilled the moment its step has been READ INTO REGISTERS (below), i.e.
//     four steps ahead of its use. (Versions one and two, NEGATIVE_RESULTS round 5: gemm_big_kernel's two-slot ring of 64-wide K-tiles, then
//     a six + three slot ring whose every step began with an exposed burst of fragment reads behind the barrier -- 0.76 and 0.82 PFLOP/s, both slower
//     than the kernel they were to replace.)
//   * the corpus plane and the query block are ALREADY in MFMA operand order (1 KiB fragments: row-block x k-block, lane = (row & 15) + 16 (k >> 3)),
//     so a DMA piece is one fragment per wave (four per wave and step), the LDS image is linear, every fragment read a conflict-free ds_read_b128;
//   * ONE barrier per step, and it certifies one step AHEAD: behind barrier S the fragments of step S + 1 have landed (counted vmcnt: the eight pieces
//     of the last two steps may still fly). So the fragment reads of step S + 1 are issued under the MFMAs of step S, whose operands were read under
//     step S - 1 (two register sets, loop unrolled by two): no step starts with an exposed LDS burst. By barrier S every wave holds step S in registers,
//     so slot S is refilled at once. The loaders never branch: past the last step they re-issue it (surplus);
//   * epilogue = the screen, every 24 steps: a lane holds, per accumulator tile, 4 consecutive queries of one row; scores >= tau (sampled k-th maximum
//     - 2B, the threshold mips_screenk32_kernel starts from) are appended to the (workgroup, query) list through an LDS counter. Lists, counts and the
//     overflow flag have the layout merge_screenk_kernel reads; no list is pruned on the way (512 slots against an expected few dozen entries: a list
//     that does run full raises the flag and the call falls through to the exact pass, as everywhere else).
// A workgroup walks row tiles b, b + G, ...: each tile's rows leave HBM once; the 384 KiB query block is re-read per tile from L2. The planes are
// allocated in multiples of 256 rows (grow()), so a tile never reads past them; rows past n_rows are masked in the epilogue.
template <bool BF>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
mips_gemmk_kernel(const char* __restrict__ Xhi, long long n_rows, const char* __restrict__ Qhi /* this group's 16 query blocks */,
                  const float* __restrict__ qbound, const float* __restrict__ tau0, int nq, u64* __restrict__ cand /* [G][kWideQ][kScreenKCap] */,
                  int* __restrict__ cand_cnt /* [G][kWideQ] */, int* __restrict__ overflow) {
    constexpr int NKB = 24;
    constexpr int SLOT = 16 * kFragBytes;         // 16 row (or query) blocks x one k-block
    constexpr int NA = 4, NQS = 4;                // ring depths = loader leads (a slot is refilled as soon as its step sits in registers); powers of two: slot = step & 3
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring_a = lds;
    char* ring_q = lds + NA * SLOT;
    int* lds_cnt = (int*)(lds + (NA + NQS) * SLOT);  // [kWideQ] entries of this workgroup's lists
    float* lds_tau = (float*)(lds_cnt + kWideQ);     // [kWideQ] thresholds (+inf for padding queries)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;       // 2 (rows) x 4 (queries)
    const int g = lane >> 4, lr = lane & 15;
    const int G = gridDim.x, b = blockIdx.x;
    const int n_tiles = (int)((n_rows + 255) / 256);
    const int n_my = b < n_tiles ? (n_tiles - b + G - 1) / G : 0;
    const int total = n_my * NKB;                  // steps of this workgroup (even)
    if (tid < kWideQ) {
        lds_cnt[tid] = 0;
        lds_tau[tid] = tid < nq ? tau0[tid] - 2.f * qbound[tid] : INFINITY;
    }
    if (n_my == 0) {
        if (tid < kWideQ) cand_cnt[(size_t)b * kWideQ + tid] = 0;
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // before any DMA is in flight (counted waits below)

    // ---- loaders. A step's 32 pieces (fragments) over 8 waves: this wave brings row blocks wave and 8 + wave, query blocks wave and 8 + wave. ----
    // Each loader keeps the address of ITS next step's fragment of the first block; the second block is 8 blocks = 8 NKB KiB further on.
    const char* pa = Xhi + ((size_t)((long long)b * 16 + wave) * NKB) * kFragBytes + lane * 16;  // row loader: tile 0, k-block 0
    const char* pq = Qhi + ((size_t)wave * NKB) * kFragBytes + lane * 16;                        // query loader
    int sa = 0, sq = 0;                    // steps issued so far by each loader
    int ka = 0, kq = 0;                    // k-block of the loader's next step
    const long long tile_jump = ((long long)G * 16 - 1) * NKB * (long long)kFragBytes + kFragBytes;  // from k-block 23 of a tile to k-block 0 of the next
    auto issue_a = [&]() __attribute__((always_inline)) {
        char* dst = ring_a + (sa & (NA - 1)) * SLOT + wave * kFragBytes;
        __builtin_amdgcn_global_load_lds(MDR_GPTR(pa), MDR_LPTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(MDR_GPTR(pa + 8 * NKB * kFragBytes), MDR_LPTR(dst + 8 * kFragBytes), 16, 0, 0);
        ++sa;
        const bool more = sa < total;  // past the last step the loader stays where it is: the surplus pieces re-read the last fragments
        const bool wrap = ka == NKB - 1;
        pa += more ? (wrap ? tile_jump : (long long)kFragBytes) : 0ll;
        ka = more ? (wrap ? 0 : ka + 1) : ka;
    };
    auto issue_q = [&]() __attribute__((always_inline)) {
        char* dst = ring_q + (sq & (NQS - 1)) * SLOT + wave * kFragBytes;
        __builtin_amdgcn_global_load_lds(MDR_GPTR(pq), MDR_LPTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(MDR_GPTR(pq + 8 * NKB * kFragBytes), MDR_LPTR(dst + 8 * kFragBytes), 16, 0, 0);
        ++sq;
        const bool more = sq < total;
        const bool wrap = kq == NKB - 1;
        pq += more ? (wrap ? -(long long)(NKB - 1) * kFragBytes : (long long)kFragBytes) : 0ll;
        kq = more ? (wrap ? 0 : kq + 1) : kq;
    };
    // prologue in the steady state's issue order: step t issues rows(t + NA), queries(t + NQS)
#pragma unroll
    for (int t = -NA; t < 0; ++t) {
        issue_a();
        if (t + NQS >= 0) issue_q();
    }

    f32x4 acc[8][4];
    half8 qf0[4], af0[8], qf1[4], af1[8];
    u64* my_lists = cand + (size_t)b * kWideQ * kScreenKCap;
    const unsigned cnt_base = (unsigned)(size_t)MDR_LPTR(lds_cnt);  // LDS byte address of the list counters
    const int rd_q = wc * 4 * kFragBytes + lane * 16, rd_a = wr * 8 * kFragBytes + lane * 16;
    int rs = 0;  // step whose fragments are read next
    auto read_step = [&](half8 (&qf)[4], half8 (&af)[8]) __attribute__((always_inline)) {
        const char* sq_ = ring_q + (rs & (NQS - 1)) * SLOT + rd_q;
        const char* sa_ = ring_a + (rs & (NA - 1)) * SLOT + rd_a;
#pragma unroll
        for (int n = 0; n < 4; ++n) qf[n] = *(const half8*)(sq_ + n * kFragBytes);
#pragma unroll
        for (int m = 0; m < 8; ++m) af[m] = *(const half8*)(sa_ + m * kFragBytes);
        ++rs;
    };
    // MFMAs of a step in two parts: the next step's fragment reads are issued BETWEEN them (hipcc does not see the inline-asm waits: it puts its own
    // lgkmcnt(0) in front of the first MFMA that uses a register set, and that wait must not have the next step's reads to wait for)
    auto mfma_rows = [&](auto zero_c, int m_lo, int m_hi, const half8 (&qf)[4], const half8 (&af)[8]) __attribute__((always_inline)) {
        constexpr bool Z = decltype(zero_c)::value;
#pragma unroll
        for (int m = 0; m < 8; ++m)
            if (m >= m_lo && m < m_hi) {
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = mfma16<BF>(qf[n], af[m], Z ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[m][n]);
            }
    };
    // steps 0 and 1 landed (the prologue issued rows 0, queries 0, rows 1, queries 1, ...: behind queries 1 come rows 2-3 and queries 2-3 = 8 pieces that may fly)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_step(qf0, af0);  // step 0 (the only exposed read burst of the kernel)
    int kb = 0, tile = 0;
    for (int S = 0; S < total; S += 2) {
        // ---- even step S: operands in set 0; behind the barrier step S + 1 has landed and everybody holds step S in registers ----
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_a();               // rows of step S + NA into step S's slot
        issue_q();               // queries of step S + NQS
        if (kb == 0) mfma_rows(std::true_type{}, 0, 2, qf0, af0);
        else mfma_rows(std::false_type{}, 0, 2, qf0, af0);
        __builtin_amdgcn_sched_barrier(0);
        read_step(qf1, af1);     // step S + 1, under the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        if (kb == 0) mfma_rows(std::true_type{}, 2, 8, qf0, af0);
        else mfma_rows(std::false_type{}, 2, 8, qf0, af0);
        // ---- odd step S + 1: operands in set 1 ----
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_a();
        issue_q();
        kb += 2;
        mfma_rows(std::false_type{}, 0, 2, qf1, af1);
        __builtin_amdgcn_sched_barrier(0);
        if (kb < NKB || S + 2 < total) read_step(qf0, af0);  // step S + 2 (the next tile's first step included); nothing behind the last step
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(std::false_type{}, 2, 8, qf1, af1);
        if (kb == NKB) {
            kb = 0;
            // ---- epilogue = the screen, under the loads in flight ----
            const long long m0 = ((long long)b + (long long)tile * G) * 256;
            ++tile;
            const long long row_lane = m0 + wr * 128 + lr;
            f32x4 tau4[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) tau4[nt] = *(const f32x4*)(lds_tau + wc * 64 + nt * 16 + 4 * g);
            bool any = false;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const bool valid = row_lane + mt * 16 < n_rows;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) any |= valid && acc[mt][nt][j] >= tau4[nt][j];
            }
            if (__ballot(any) != 0ull) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const long long row = row_lane + mt * 16;
                    const bool valid = row < n_rows;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const f32x4 v = acc[mt][nt];
                        unsigned m = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) m |= (valid && v[j] >= tau4[nt][j]) ? (1u << j) : 0u;
                        if (__ballot(m != 0u) == 0ull) continue;  // wave-uniform: most accumulator tiles have no hit
                        while (m) {
                            const int j = __builtin_ctz(m);
                            m &= m - 1;
                            const int q = wc * 64 + nt * 16 + 4 * g + j;
                            int pos;  // (inline asm: hipcc puts s_waitcnt vmcnt(0) in front of an LDS atomic it can see while LDS-DMA loads are in flight -- that would drain the ring on every hit)
                            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(cnt_base + 4u * (unsigned)q), "v"(1) : "memory");
                            if (pos < kScreenKCap) my_lists[(size_t)q * kScreenKCap + pos] = make_key(v[j], (unsigned)row);
                            else *overflow = 1;
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus pieces must have landed before the LDS is released
    __syncthreads();
    if (tid < kWideQ) {
        const int c = lds_cnt[tid];
        cand_cnt[(size_t)b * kWideQ + tid] = c < kScreenKCap ? c : kScreenKCap;
    }
}

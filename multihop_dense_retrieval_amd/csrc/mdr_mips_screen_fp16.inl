// csrc/mdr_mips_screen_fp16.inl -- the fp16 hi-plane screens: k = 1 (16 and 32 queries per wave), exact re-scoring (exact_dot16, refine), 2 <= k <= 128 (screen-k, 16 and 32
// queries per wave) and its merge kernel. Included by mdr_mips.hip inside namespace mdr::{anonymous}.
// ---- the screen kernel (k == 1): hi plane only, one MFMA per k-block -------------------------------------
// Stage = one "super-block" of 32 rows (2 row-blocks) of the hi plane = 2*NKB KiB; 3-slot ring as above.
// Iteration 0 only samples (publishes the largest s_hi to gmax, emits nothing) so that the cut is tight
// before candidates are emitted; its super-block is revisited as the last iteration.
template <int NKB>
__device__ __forceinline__ void issue_super_block(const char* __restrict__ Xhi, int sb, char* slot, int wave, int lane) {
    constexpr int CPW = NKB / 4;  // 2*NKB pieces over 8 waves
    const char* g = Xhi + ((size_t)sb * 2 * NKB + (size_t)wave * CPW) * kFragBytes + lane * 16;
    char* l = slot + wave * CPW * kFragBytes;
#pragma unroll
    for (int c = 0; c < CPW; ++c) __builtin_amdgcn_global_load_lds(MDR_GPTR(g + c * kFragBytes), MDR_LPTR(l + c * kFragBytes), 16, 0, MDR_MIPS_DMA_AUX);
}

__device__ __forceinline__ unsigned load_u32_l2(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kWaveCandCap = 2048;  // (query,row) candidates one wave may emit per pass before the exact fallback
constexpr int kSampleStages = 4;    // super-blocks per workgroup the sample pass scores (x 256 workgroups x 32 rows)
constexpr int kSampleStagesK = 16;  // same for the k > 1 sample pass (MODE 2): 131k rows, spread over each workgroup's range

// MODE 0 (sample pass): score the first kSampleStages super-blocks of every workgroup, emit nothing, publish
//         the largest s_hi per query to gmax. The kernel boundary is the grid-wide synchronisation.
// MODE 1 (main pass):   start from gmax, score every row, append rows with s_hi >= known - 2B to this wave's
//         PRIVATE candidate list (slot from a ballot prefix: no returning atomic, so nothing ever waits on
//         vmcnt and the corpus DMA is never drained), tighten `known` with the wave's own maxima.
template <bool BF>
__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int NKB, int MODE, bool BF>
__global__ void __launch_bounds__(512, 2)
mips_screen_kernel(const char* __restrict__ Xhi, long long n_rows, int n_sb, const char* __restrict__ Qhi, const float* __restrict__ qbound, int nq,
                   int q_base, unsigned* __restrict__ gmax /* [nq] ordered(max s_hi) */, u64* __restrict__ cand /* [waves][kWaveCandCap] */,
                   int* __restrict__ cand_cnt /* [waves] */, int* __restrict__ overflow, const int* __restrict__ run_if = nullptr,
                   int sample_stages = kSampleStagesK /* MODE 2: stages per workgroup; gmax is [G * sample_stages][kStreamQ], one maximum per stage */) {
    if (run_if && *run_if == 0) return;  // behind the int8 tier: only when one of its lists overflowed
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB_BYTES = 2 * NKB * kFragBytes;
    constexpr int CPW = NKB / 4;
    constexpr int HK = NKB / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    int n_it = (n_sb - b + G - 1) / G;  // >= 1 (grid <= n_sb)
    if (MODE == 0 && n_it > kSampleStages) n_it = kSampleStages;
    int step_ = 1;  // MODE 2 spreads its sample stages over the workgroup's whole row range
    if (MODE == 2 && n_it > sample_stages) { step_ = n_it / sample_stages; n_it = sample_stages; }
    const int sG = (MODE == 2 ? step_ : 1) * G;  // super-block stride between consecutive stages

    issue_super_block<NKB>(Xhi, b, lds, wave, lane);
    if (n_it > 1) issue_super_block<NKB>(Xhi, b + sG, lds + SB_BYTES, wave, lane);

    const bool wave_active = wave * 16 < nq;
    half8 qh[NKB];
    {
        const size_t qoff = (size_t)wave * NKB * kFragBytes + lane * 16;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) qh[kb] = *(const half8*)(Qhi + qoff + kb * kFragBytes);
    }
    const int qlocal = wave * 16 + (lane & 15);
    const bool q_valid = qlocal < nq;
    float band2 = 0.f;
    float known = -FLT_MAX;  // largest s_hi known for this lane's query (sample pass + this wave's own rows)
    if (MODE == 1 && q_valid) {
        band2 = 2.f * qbound[qlocal];
        unsigned g = gmax[qlocal];
        if (g) known = unord32(g);
    }
    // retire every load above before the loop: pending VMEM results would make the compiler put vmcnt(0) in front
    // of their first use inside the loop and drain the corpus DMA each iteration
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(qh[kb]));
    asm volatile("" : "+v"(band2), "+v"(known));
    const unsigned sub_row = 4u * (unsigned)(lane >> 4);
    float hmax = -FLT_MAX;  // largest s_hi this lane has seen
    int my_cnt = 0;         // wave-uniform: entries in this wave's candidate list
    u64* my_list = cand + ((size_t)b * 8 + wave) * kWaveCandCap;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 1 && ((it + b) & 63) == 63 && wave_active) {
            // Every 64 stages (staggered over the workgroups so gmax is not hammered by all of them at once) exchange
            // maxima with the other workgroups. The load makes the compiler wait vmcnt(0); placed HERE, before this
            // iteration's DMA is issued, the only VMEM ops outstanding are last iteration's (already landed) pieces.
            float hm = fmaxf(hmax, __shfl_xor(hmax, 16));
            hm = fmaxf(hm, __shfl_xor(hm, 32));
            float kn = known;
            if (lane < 16 && q_valid) {
                if (hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
                unsigned g = load_u32_l2(gmax + qlocal);
                if (g) kn = fmaxf(kn, unord32(g));
            }
            known = __shfl(kn, lane & 15);
        }
        if (it + 2 < n_it) issue_super_block<NKB>(Xhi, b + (it + 2) * sG, lds + ((it + 2) % 3) * SB_BYTES, wave, lane);
        if (!wave_active) continue;

        const char* p = lds + (it % 3) * SB_BYTES + lane * 16;
        // 4 independent accumulation chains: {row-block 0, 1} x {first, second half of K}; LDS reads run
        // PF k-steps ahead of the MFMAs that consume them (issue order pinned below)
        f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;
        constexpr int PF = 2;
        half8 x00[PF], x01[PF], x10[PF], x11[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            x00[i] = *(const half8*)(p + i * kFragBytes);
            x01[i] = *(const half8*)(p + (HK + i) * kFragBytes);
            x10[i] = *(const half8*)(p + (NKB + i) * kFragBytes);
            x11[i] = *(const half8*)(p + (NKB + HK + i) * kFragBytes);
        }
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) {
            const half8 c00 = x00[kb % PF], c01 = x01[kb % PF], c10 = x10[kb % PF], c11 = x11[kb % PF];
            if (kb + PF < HK) {
                x00[kb % PF] = *(const half8*)(p + (kb + PF) * kFragBytes);
                x01[kb % PF] = *(const half8*)(p + (HK + kb + PF) * kFragBytes);
                x10[kb % PF] = *(const half8*)(p + (NKB + kb + PF) * kFragBytes);
                x11[kb % PF] = *(const half8*)(p + (NKB + HK + kb + PF) * kFragBytes);
            }
            a00 = mfma16<BF>(c00, qh[kb], a00);
            a01 = mfma16<BF>(c01, qh[HK + kb], a01);
            a10 = mfma16<BF>(c10, qh[kb], a10);
            a11 = mfma16<BF>(c11, qh[HK + kb], a11);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * PF, 0);
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) {
            if (kb + PF < HK) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        const f32x4 s0 = a00 + a01, s1 = a10 + a11;
        const unsigned row0 = (unsigned)(b + it * sG) * 32u + sub_row;
        const float cut = known - band2;  // a row below this cannot beat the row that produced `known`
        // Fast path (almost every stage): the super-block lies inside the corpus and no score reaches the cut -> 7 max
        // operations and ONE ballot instead of 8 compare / ballot / branch sequences.
        const bool whole = (long long)(b + it * sG) * 32 + 32 <= n_rows;  // wave-uniform
        const float m8 = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        if (whole && (MODE != 1 || __ballot(q_valid && m8 >= cut) == 0ull)) {
            if (q_valid) hmax = fmaxf(hmax, m8);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = h ? s1[r] : s0[r];
                    const unsigned row = row0 + 16u * h + r;
                    const bool ok = (long long)row < n_rows && q_valid;
                    if (ok) hmax = fmaxf(hmax, sc);
                    if (MODE == 1) {
                        const bool hit = ok && sc >= cut;
                        const u64 m = __ballot(hit);
                        if (m) {  // wave-uniform
                            const int slot = my_cnt + __popcll(m & lt);
                            if (hit && slot < kWaveCandCap) my_list[slot] = ((u64)(unsigned)(q_base + qlocal) << 32) | row;
                            my_cnt += __popcll(m);
                        }
                    }
                }
        }
        if (MODE == 1) {  // share the maximum between the 4 lanes of a query
            float hm = fmaxf(hmax, __shfl_xor(hmax, 16));
            hm = fmaxf(hm, __shfl_xor(hm, 32));
            known = fmaxf(known, hm);
        }
        if (MODE == 2) {  // one maximum per (workgroup, stage, query): 32 disjoint rows each, so the k-th largest of them is a bound on the k-th best row
            float hm = fmaxf(hmax, __shfl_xor(hmax, 16));
            hm = fmaxf(hm, __shfl_xor(hm, 32));
            if (lane < 16 && q_valid) gmax[((size_t)b * sample_stages + it) * kStreamQ + qlocal] = hm > -FLT_MAX ? ord32(hm) : 0u;
            hmax = -FLT_MAX;
        }
    }
    if (MODE == 0) {
        float hm = fmaxf(hmax, __shfl_xor(hmax, 16));
        hm = fmaxf(hm, __shfl_xor(hm, 32));
        if (lane < 16 && q_valid && hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
    } else if (MODE == 2) {  // (published per stage, above; the host zeroes the table first: a workgroup with fewer stages leaves zeros = "no row")
    } else if (lane == 0) {
        cand_cnt[b * 8 + wave] = my_cnt < kWaveCandCap ? my_cnt : kWaveCandCap;
        if (my_cnt > kWaveCandCap) *overflow = 1;
    }
}

// ---- the screen kernel with 32 queries per wave (256 per pass): v_mfma_f32_32x32x16 -------------------------------------
// Same streaming skeleton, same bound, same candidate lists and the same two passes (MODE 0 sample / MODE 1 main, MODE 2
// per-workgroup maxima for k > 1) as mips_screen_kernel; what changes is the tile: one 32x32x16 MFMA multiplies the WHOLE
// 32-row super-block with 32 queries, so a wave keeps 32 queries resident (48 K-slices x 4 VGPRs = 192 registers) and a
// corpus pass serves 256 queries instead of 128. Used when a call brings more than 128 queries (hop 2 at beam >= 2, the
// weak-scaling bench, and the hop-2 + next hop-1 searches of the pipelined loop): half the corpus passes.
// The stored corpus layout (16-row fragment blocks for 16x16x32) is read with a different address pattern: K-slice s of a
// 32-row super-block, lane (row = l & 31, k = 16 s + 8 (l >> 5) ..) sits at
//     ((l >> 4) & 1) * NKB KiB  +  (s >> 1) KiB  +  (s & 1) * 512  +  (l >> 5) * 256  +  (l & 15) * 16
// of the super-block image: 16 consecutive 16-B slots per ds_read_b128 lane group, conflict-free.
// Accumulator layout (32x32): lane holds query l & 31 and corpus rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWideQ = 256;  // queries per pass of the 32-queries-per-wave kernels

template <bool BF>
__device__ __forceinline__ f32x16 mfma32(half8 a, half8 b, f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// The 2*NKB-step accumulate chain of one 32-row super-block: acc = sum_s A_s(LDS) x qf[s]. hipcc waits lgkmcnt(0) in front of
// every group of four MFMAs here (all reads in flight, the newest issued one instruction earlier: ~100 exposed cycles per 128
// of matrix work), so the LDS reads and their waits are hand-placed: PF reads in flight, `s_waitcnt lgkmcnt(PF-1)` retires
// exactly the oldest one before the MFMA that consumes it, the freed registers are refilled at once. The MFMAs stay compiler
// builtins (hazards and accumulator allocation are hipcc's); each wait names the fragment it retires as an in/out operand, which
// pins MFMA s behind wait s, and the refill behind MFMA s (guide §5.7, form ii). No other LDS / scalar-memory operation may
// sit inside this region (checked in the .s: none), else the counts would be off.
template <int NKB, bool BF>
__device__ __forceinline__ f32x16 mfma_chain32(const char* p, const half8 (&qf)[2 * NKB]) {
    constexpr int NS = 2 * NKB, PF = 4;  // (2, 4, 6 reads in flight measured the same: 2.11 / 2.07 / 2.07 ms at nq = 256)
    const unsigned a = (unsigned)(uintptr_t)p;  // LDS byte address (low 32 bits of the flat pointer)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    half8 xa[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[i]) : "v"(a), "n"((i >> 1) * kFragBytes + (i & 1) * 512));
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int left = NS - 1 - sl < PF - 1 ? NS - 1 - sl : PF - 1;  // reads younger than the one needed now
        switch (left) {
            case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(xa[sl % PF])); break;
            case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xa[sl % PF])); break;
            case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(xa[sl % PF])); break;
            case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xa[sl % PF])); break;
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xa[sl % PF])); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xa[sl % PF])); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xa[sl % PF])); break;
            default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[sl % PF])); break;
        }
        acc = mfma32<BF>(xa[sl % PF], qf[sl], acc);
        if (sl + PF < NS)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[sl % PF]) : "v"(a), "n"(((sl + PF) >> 1) * kFragBytes + ((sl + PF) & 1) * 512));
    }
    return acc;
}

template <int NKB, int MODE, bool BF>
__global__ void __launch_bounds__(512, 2)
mips_screen32_kernel(const char* __restrict__ Xhi, long long n_rows, int n_sb, const char* __restrict__ Qhi, const float* __restrict__ qbound, int nq,
                     int q_base, unsigned* __restrict__ gmax /* [nq] ordered(max s_hi); MODE 2: [G][kWideQ] */,
                     u64* __restrict__ cand /* [waves][kWaveCandCap] */, int* __restrict__ cand_cnt /* [waves] */, int* __restrict__ overflow,
                     const int* __restrict__ run_if = nullptr, int sample_stages = kSampleStagesK /* MODE 2: gmax is [G * sample_stages][kWideQ] */) {
    if (run_if && *run_if == 0) return;  // behind the int8 tier: only when one of its lists overflowed
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB_BYTES = 2 * NKB * kFragBytes;
    constexpr int CPW = NKB / 4;
    constexpr int NS = 2 * NKB;  // 16-deep K slices
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    int n_it = (n_sb - b + G - 1) / G;  // >= 1 (grid <= n_sb)
    if (MODE == 0 && n_it > kSampleStages) n_it = kSampleStages;
    int step_ = 1;
    if (MODE == 2 && n_it > sample_stages) { step_ = n_it / sample_stages; n_it = sample_stages; }
    const int sG = (MODE == 2 ? step_ : 1) * G;

    issue_super_block<NKB>(Xhi, b, lds, wave, lane);
    if (n_it > 1) issue_super_block<NKB>(Xhi, b + sG, lds + SB_BYTES, wave, lane);

    const bool wave_active = wave * 32 < nq;
    const int l31 = lane & 31, lh = lane >> 5;
    half8 qf[NS];
    {
        // query row 32 w + l31 of the fragment-tiled query matrix (16-row blocks): K-slice s -> 16-B chunk 2 s + lh
        const size_t qrow = (size_t)wave * 32 + l31;
        const char* qp = Qhi + (qrow >> 4) * ((size_t)NKB * kFragBytes) + (qrow & 15) * 16 + lh * 256;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) qf[sl] = *(const half8*)(qp + (sl >> 1) * kFragBytes + (sl & 1) * 512);
    }
    const int qlocal = wave * 32 + l31;
    const bool q_valid = qlocal < nq;
    float band2 = 0.f;
    float known = -FLT_MAX;
    if (MODE == 1 && q_valid) {
        band2 = 2.f * qbound[qlocal];
        unsigned g = gmax[qlocal];
        if (g) known = unord32(g);
    }
    // retire every load above before the loop (see mips_screen_kernel)
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) asm volatile("" : "+v"(qf[sl]));
    asm volatile("" : "+v"(band2), "+v"(known));
    float hmax = -FLT_MAX;
    int my_cnt = 0;  // wave-uniform
    u64* my_list = cand + ((size_t)b * 8 + wave) * kWaveCandCap;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int rd_off = ((lane >> 4) & 1) * (NKB * kFragBytes) + lh * 256 + (lane & 15) * 16;

    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 1 && ((it + b) & 63) == 63 && wave_active) {  // exchange maxima with the other workgroups (see mips_screen_kernel)
            float hm = fmaxf(hmax, __shfl_xor(hmax, 32));
            float kn = known;
            if (lane < 32 && q_valid) {
                if (hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
                unsigned g = load_u32_l2(gmax + qlocal);
                if (g) kn = fmaxf(kn, unord32(g));
            }
            known = __shfl(kn, l31);
        }
        if (it + 2 < n_it) issue_super_block<NKB>(Xhi, b + (it + 2) * sG, lds + ((it + 2) % 3) * SB_BYTES, wave, lane);
        if (!wave_active) continue;

        const char* p = lds + (it % 3) * SB_BYTES + rd_off;
        const f32x16 acc = mfma_chain32<NKB, BF>(p, qf);
        const unsigned row0 = (unsigned)(b + it * sG) * 32u + 4u * (unsigned)lh;
        const float cut = known - band2;
        // Fast path: a super-block that lies completely inside the corpus and holds no score above the cut (almost all of
        // them) costs 15 max operations and ONE ballot instead of 16 compare / ballot / branch sequences.
        const bool whole = (long long)(b + it * sG) * 32 + 32 <= n_rows;  // wave-uniform
        float m16 = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m16 = fmaxf(m16, acc[r]);
        if (whole && (MODE != 1 || __ballot(q_valid && m16 >= cut) == 0ull)) {
            if (q_valid) hmax = fmaxf(hmax, m16);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = acc[r];
                const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
                const bool ok = (long long)row < n_rows && q_valid;
                if (ok) hmax = fmaxf(hmax, sc);
                if (MODE == 1) {
                    const bool hit = ok && sc >= cut;
                    const u64 m = __ballot(hit);
                    if (m) {  // wave-uniform
                        const int slot = my_cnt + __popcll(m & lt);
                        if (hit && slot < kWaveCandCap) my_list[slot] = ((u64)(unsigned)(q_base + qlocal) << 32) | row;
                        my_cnt += __popcll(m);
                    }
                }
            }
        }
        if (MODE == 1) known = fmaxf(known, fmaxf(hmax, __shfl_xor(hmax, 32)));  // the two lanes of a query share their maxima
        if (MODE == 2) {  // one maximum per (workgroup, stage, query), see mips_screen_kernel
            const float hm = fmaxf(hmax, __shfl_xor(hmax, 32));
            if (lane < 32 && q_valid) gmax[((size_t)b * sample_stages + it) * kWideQ + qlocal] = hm > -FLT_MAX ? ord32(hm) : 0u;
            hmax = -FLT_MAX;
        }
    }
    if (MODE == 0) {
        const float hm = fmaxf(hmax, __shfl_xor(hmax, 32));
        if (lane < 32 && q_valid && hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
    } else if (MODE == 2) {  // (published per stage, above)
    } else if (lane == 0) {
        cand_cnt[b * 8 + wave] = my_cnt < kWaveCandCap ? my_cnt : kWaveCandCap;
        if (my_cnt > kWaveCandCap) *overflow = 1;
    }
}

// exact fp32 score of one (query, row) pair by a 16-lane group (sub = lane within the group): FMA over the
// reconstructed values of both planes, then a 16-lane butterfly. Every lane of the group returns the sum.
template <bool BF>
__device__ __forceinline__ float exact_dot16(const char* __restrict__ Xhi, const char* __restrict__ Xlo, int nkb, const float* __restrict__ qrow,
                                             unsigned row, int sub, float xs /* 2^E: stored rows -> the caller's scale (exact) */) {
    const size_t base = ((size_t)(row >> 4) * nkb) * kFragBytes + (size_t)(row & 15) * 16;
    float acc = 0.f;
    for (int pc = sub; pc < nkb * 4; pc += 16) {  // piece = (k-block, 8-column group)
        const int kb = pc >> 2, g = pc & 3;
        const size_t off = base + (size_t)kb * kFragBytes + (size_t)g * 256;
        const float* qp = qrow + kb * 32 + g * 8;
        if (BF) {
            const ushort8 hb = *(const ushort8*)(Xhi + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(bf16_bits_to_f32(hb[j]), qp[j], acc);
        } else {
            const half8 h = *(const half8*)(Xhi + off);
            const half8 l = *(const half8*)(Xlo + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf((float)h[j] + (float)l[j] * kLoInv, qp[j], acc);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc * xs;
}

// exact re-scoring of the screen kernel's candidates: 16 lanes per (query, row);
// one 256-thread block per source wave list, 16 candidates in flight per block
template <bool BF>
__global__ void __launch_bounds__(256)
mips_refine_kernel(const char* __restrict__ Xhi, const char* __restrict__ Xlo, int nkb, const float* __restrict__ q, const u64* __restrict__ cand,
                   const int* __restrict__ cand_cnt, u64* __restrict__ best, float xs, const int* __restrict__ run_if = nullptr) {
    if (run_if && *run_if == 0) return;
    const int n = cand_cnt[blockIdx.x];
    if (n == 0) return;
    const u64* list = cand + (size_t)blockIdx.x * kWaveCandCap;
    const int sub = threadIdx.x & 15;  // lane within the 16-lane group
    const int d = nkb * 32;
    for (int c = threadIdx.x >> 4; c < n; c += 16) {
        const u64 e = list[c];
        const unsigned qi = (unsigned)(e >> 32), row = (unsigned)e;
        const float acc = exact_dot16<BF>(Xhi, Xlo, nkb, q + (size_t)qi * d, row, sub, xs);
        if (sub == 0) atomicMax(best + qi, make_key(acc, row));
    }
}

// ---- the screen kernel for 2 <= k <= 256: hi plane only, per-(workgroup, query) lists keyed by s_hi ---------
// A row can be among the k best exact scores only if s_hi >= h_k - 2B, h_k = k-th largest s_hi over ALL rows
// (k rows have exact >= h_k - B, and s_hi < h_k - 2B means exact < h_k - B). Any lower bound on h_k will do:
//   1. sample pass (mips_screen_kernel MODE 2): every workgroup scores kSampleStagesK (k > 128: twice as many) super-blocks and publishes the
//      largest s_hi of EACH per query; the k-th largest of those G x stages maxima (k distinct rows!) is the first bound (tau0). (Rounds 1-3 kept one
//      maximum per workgroup: at k = 250 the 250th of 256 maxima is a bound ~36 k rows deep per query at 5 M rows; per-stage maxima of a 262 k-row
//      sample put it ~5 k deep, inside what the merge kernel holds, which is what lets the screen path take k up to 256.)
//   2. main pass: rows with s_hi >= bound - 2B are appended to the (workgroup, query) list; should a list run full it
//      is pruned to (its own k-th largest) - 2B, which becomes that list's bound. Bounds only rise and never exceed
//      h_k, so the union of the lists holds every possible winner.
//   3. merge_screenk_kernel: h_k over the union, keep the band, re-score it exactly (both planes), sort.
constexpr int kScreenKCap = 512;       // slots per (workgroup, query)
constexpr int kSurvMax = 1024;         // band survivors per query the merge kernel re-scores before giving up (-> exact fallback)
constexpr int kMergeKLds = 15360;      // union keys per query the merge kernel holds in LDS (120 KiB) before giving up

__device__ __forceinline__ u64 floor_key(float score) { return (u64)ord32(score) << 32; }  // smallest key with that score

// Whole wave, identical arguments, c >= k. list[0..c): unique keys. Finds t = k-th largest key, keeps the keys with
// score >= score(t) - band2 compacted to the front. If that band would leave fewer than 32 free slots it keeps only
// the k best and raises *overflow (the results of this pass are then discarded by the exact fallback). Returns t.
template <int E>
__device__ inline u64 wave_select_band(u64* list, int c, int k, float band2, int lane, int* new_count, int* overflow) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's appends have reached L2
    u64 key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int idx = e * 64 + lane;
        key[e] = idx < c ? load_key_l2(list + idx) : 0ull;
    }
    u64 t = 0ull;
    for (int bit = 63; bit >= 0; --bit) {
        u64 cnd = t | (1ull << bit);
        int n = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) n += __popcll(__ballot(key[e] >= cnd));
        if (n >= k) t = cnd;
    }
    u64 cut = floor_key(key_score(t) - band2);
    int n_band = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) n_band += __popcll(__ballot(key[e] >= cut && key[e] != 0ull));
    if (n_band > E * 64 - 64) {
        cut = t;
        if (lane == 0) *overflow = 1;
    }
    int base = 0;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int e = 0; e < E; ++e) {
        bool p = key[e] >= cut && key[e] != 0ull;
        u64 m = __ballot(p);
        if (p) list[base + __popcll(m & lt)] = key[e];
        base += __popcll(m);
    }
    *new_count = base;
    return t;
}

// k-th largest of the V = G x stages sample maxima of each query (one per workgroup and sample stage: disjoint 32-row blocks, i.e. distinct rows) -> tau0
// (-inf when fewer than k blocks saw a row). Bisection on the ordered 32-bit key: V is 4 k - 32 k values, read from L2 each step.
__global__ void __launch_bounds__(256) kth_of_maxima_kernel(const unsigned* __restrict__ wgmax /* [V][qcap] ordered, 0 = none */, int V, int k,
                                                            float* __restrict__ tau0 /* [qcap] */, int qcap) {
    const int ql = blockIdx.x;
    __shared__ int red[4];
    unsigned t = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned c = t | (1u << bit);
        int n = 0;
        for (int i = threadIdx.x; i < V; i += 256) n += wgmax[(size_t)i * qcap + ql] >= c;
        n = block_sum_256(n, red);
        if (n >= k) t = c;
    }
    if (threadIdx.x == 0) tau0[ql] = t ? unord32(t) : -INFINITY;  // t == 0: fewer than k non-empty blocks
}

template <int NKB, bool BF>
__global__ void __launch_bounds__(512, 2)
mips_screenk_kernel(const char* __restrict__ Xhi, long long n_rows, int n_sb, const char* __restrict__ Qhi, const float* __restrict__ qbound,
                    const float* __restrict__ tau0, int nq, u64* __restrict__ cand /* [G][kStreamQ][kScreenKCap] */, int* __restrict__ cand_cnt, int k,
                    int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB_BYTES = 2 * NKB * kFragBytes;
    constexpr int CPW = NKB / 4;
    constexpr int HK = NKB / 2;
    constexpr int E = kScreenKCap / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    const int n_it = (n_sb - b + G - 1) / G;  // >= 1 (grid <= n_sb)

    issue_super_block<NKB>(Xhi, b, lds, wave, lane);
    if (n_it > 1) issue_super_block<NKB>(Xhi, b + G, lds + SB_BYTES, wave, lane);

    const bool wave_active = wave * 16 < nq;
    half8 qh[NKB];
    {
        const size_t qoff = (size_t)wave * NKB * kFragBytes + lane * 16;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) qh[kb] = *(const half8*)(Qhi + qoff + kb * kFragBytes);
    }
    const int qlocal = wave * 16 + (lane & 15);
    const bool q_valid = qlocal < nq;
    float band2 = q_valid ? 2.f * qbound[qlocal] : 0.f;
    float tau = q_valid ? tau0[qlocal] - band2 : INFINITY;  // rows below this s_hi cannot be among the k best of this lane's query
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(qh[kb]));  // retire the loads before the DMA loop (see mips_stream_kernel)
    asm volatile("" : "+v"(band2), "+v"(tau));
    const unsigned sub_row = 4u * (unsigned)(lane >> 4);
    u64* wave_lists = cand + ((size_t)b * kStreamQ + (size_t)wave * 16) * kScreenKCap;
    u64* my_list = wave_lists + (size_t)(lane & 15) * kScreenKCap;
    int cnt = 0;  // entries in this lane's query list; replicated in the 4 lanes (l, l^16, l^32, l^48) that share the query
    const u64 below_mask = (1ull << (lane & 48)) - 1ull;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;  // scores of the previous stage, consumed one iteration later
    unsigned prow0 = 0;

    // Appends of stage it-1 are issued at the top of iteration it, BEFORE that iteration's DMA: the stores are then older
    // than the newest DMA batch and the counted vmcnt wait of the next iteration does not have to cover that batch.
    // Slots come from a ballot prefix over the 4 lanes of a query: no atomics, no LDS, nothing that waits on vmcnt.
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sc = j < 4 ? s0[j & 3] : s1[j & 3];
            const unsigned row = prow0 + 16u * (j >> 2) + (j & 3);
            const bool hit = q_valid && (long long)row < n_rows && sc >= tau;
            const u64 m = __ballot(hit);
            if (m) {  // wave-uniform
                const u64 grp = (m >> (lane & 15)) & 0x0001000100010001ull;
                const int slot = cnt + __popcll(grp & below_mask);
                if (hit && slot < kScreenKCap) my_list[slot] = make_key(sc, row);
                cnt += __popcll(grp);
            }
        }
        unsigned m16 = (unsigned)(__ballot(cnt > kScreenKCap - 32) & 0xFFFFull);
        while (m16) {  // rare: a list is about to run full -> prune it to (its k-th largest) - 2B
            const int qi = __builtin_ctz(m16);
            m16 &= m16 - 1;
            int nc;
            const u64 t = wave_select_band<E>(wave_lists + (size_t)qi * kScreenKCap, __shfl(cnt, qi), k, __shfl(band2, qi), lane, &nc, overflow);
            if ((lane & 15) == qi) {
                cnt = nc;
                tau = fmaxf(tau, key_score(t) - band2);
            }
        }
    };

    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wave_active && it > 0) flush();
        if (it + 2 < n_it) issue_super_block<NKB>(Xhi, b + (it + 2) * G, lds + ((it + 2) % 3) * SB_BYTES, wave, lane);
        if (!wave_active) continue;

        const char* p = lds + (it % 3) * SB_BYTES + lane * 16;
        f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;
        constexpr int PF = 2;
        half8 x00[PF], x01[PF], x10[PF], x11[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            x00[i] = *(const half8*)(p + i * kFragBytes);
            x01[i] = *(const half8*)(p + (HK + i) * kFragBytes);
            x10[i] = *(const half8*)(p + (NKB + i) * kFragBytes);
            x11[i] = *(const half8*)(p + (NKB + HK + i) * kFragBytes);
        }
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) {
            const half8 c00 = x00[kb % PF], c01 = x01[kb % PF], c10 = x10[kb % PF], c11 = x11[kb % PF];
            if (kb + PF < HK) {
                x00[kb % PF] = *(const half8*)(p + (kb + PF) * kFragBytes);
                x01[kb % PF] = *(const half8*)(p + (HK + kb + PF) * kFragBytes);
                x10[kb % PF] = *(const half8*)(p + (NKB + kb + PF) * kFragBytes);
                x11[kb % PF] = *(const half8*)(p + (NKB + HK + kb + PF) * kFragBytes);
            }
            a00 = mfma16<BF>(c00, qh[kb], a00);
            a01 = mfma16<BF>(c01, qh[HK + kb], a01);
            a10 = mfma16<BF>(c10, qh[kb], a10);
            a11 = mfma16<BF>(c11, qh[HK + kb], a11);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * PF, 0);
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) {
            if (kb + PF < HK) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        s0 = a00 + a01;
        s1 = a10 + a11;
        prow0 = (unsigned)(b + it * G) * 32u + sub_row;
    }
    if (wave_active) {
        flush();
        if (lane < 16) cand_cnt[(size_t)b * kStreamQ + qlocal] = cnt;
    }
}

// ---- the screen-k kernel with 32 queries per wave (256 per pass): see mips_screen32_kernel for the tile and mips_screenk_kernel
// for the list protocol. Two lanes (l, l + 32) share a query; a stage's appends are issued at once (the one-ballot fast path makes
// stages with a hit the exception, so the stores rarely sit between the DMA batches of the counted vmcnt wait).
#ifndef MDR_SK32_ABL
#define MDR_SK32_ABL 0  // measurement builds (WRONG results): 1 = no DMA behind the prologue (compute on stale LDS), 2 = DMA only (no MFMA chain, no lists)
#endif
template <int NKB, bool BF>
__global__ void __launch_bounds__(512, 2)
mips_screenk32_kernel(const char* __restrict__ Xhi, long long n_rows, int n_sb, const char* __restrict__ Qhi, const float* __restrict__ qbound,
                      const float* __restrict__ tau0, int nq, u64* __restrict__ cand /* [G][kWideQ][kScreenKCap] */, int* __restrict__ cand_cnt, int k,
                      int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB_BYTES = 2 * NKB * kFragBytes;
    constexpr int CPW = NKB / 4;
    constexpr int NS = 2 * NKB;
    constexpr int E = kScreenKCap / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    const int n_it = (n_sb - b + G - 1) / G;  // >= 1 (grid <= n_sb)

    issue_super_block<NKB>(Xhi, b, lds, wave, lane);
    if (n_it > 1) issue_super_block<NKB>(Xhi, b + G, lds + SB_BYTES, wave, lane);

    const bool wave_active = wave * 32 < nq;
    const int l31 = lane & 31, lh = lane >> 5;
    half8 qf[NS];
    {
        const size_t qrow = (size_t)wave * 32 + l31;
        const char* qp = Qhi + (qrow >> 4) * ((size_t)NKB * kFragBytes) + (qrow & 15) * 16 + lh * 256;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) qf[sl] = *(const half8*)(qp + (sl >> 1) * kFragBytes + (sl & 1) * 512);
    }
    const int qlocal = wave * 32 + l31;
    const bool q_valid = qlocal < nq;
    float band2 = q_valid ? 2.f * qbound[qlocal] : 0.f;
    float tau = q_valid ? tau0[qlocal] - band2 : INFINITY;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) asm volatile("" : "+v"(qf[sl]));
    asm volatile("" : "+v"(band2), "+v"(tau));
    u64* wave_lists = cand + ((size_t)b * kWideQ + (size_t)wave * 32) * kScreenKCap;
    u64* my_list = wave_lists + (size_t)l31 * kScreenKCap;
    int cnt = 0;  // entries in this lane's query list; replicated in the two lanes that share the query
    const int rd_off = ((lane >> 4) & 1) * (NKB * kFragBytes) + lh * 256 + (lane & 15) * 16;

    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 2 < n_it && MDR_SK32_ABL != 1) issue_super_block<NKB>(Xhi, b + (it + 2) * G, lds + ((it + 2) % 3) * SB_BYTES, wave, lane);
        if (!wave_active || MDR_SK32_ABL == 2) continue;

        const char* p = lds + (MDR_SK32_ABL == 1 ? (it & 1) : (it % 3)) * SB_BYTES + rd_off;
        const f32x16 acc = mfma_chain32<NKB, BF>(p, qf);
        float m16 = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m16 = fmaxf(m16, acc[r]);
        if (MDR_SK32_ABL) { asm volatile("" : "+v"(m16)); continue; }
        if (__ballot(q_valid && m16 >= tau) == 0ull) continue;  // nothing of this super-block can enter any list (tau = +inf for padding lanes)
        const unsigned row0 = (unsigned)(b + it * G) * 32u + 4u * (unsigned)lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sc = acc[r];
            const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
            const bool hit = q_valid && (long long)row < n_rows && sc >= tau;
            const u64 m = __ballot(hit);
            if (m) {  // wave-uniform
                const u64 grp = (m >> l31) & 0x0000000100000001ull;  // the two lanes of this query
                const int slot = cnt + (lh ? (int)(grp & 1ull) : 0);
                if (hit && slot < kScreenKCap) my_list[slot] = make_key(sc, row);
                cnt += __popcll(grp);
            }
        }
        unsigned m32 = (unsigned)(__ballot(cnt > kScreenKCap - 32) & 0xFFFFFFFFull);
        while (m32) {  // rare: a list is about to run full -> prune it to (its k-th largest) - 2B
            const int qi = __builtin_ctz(m32);
            m32 &= m32 - 1;
            int nc;
            const u64 t = wave_select_band<E>(wave_lists + (size_t)qi * kScreenKCap, __shfl(cnt, qi), k, __shfl(band2, qi), lane, &nc, overflow);
            if (l31 == qi) {
                cnt = nc;
                tau = fmaxf(tau, key_score(t) - band2);
            }
        }
    }
    if (wave_active && lane < 32) cand_cnt[(size_t)b * kWideQ + qlocal] = cnt;
}

// One 256-thread block per query: union of the G lists -> h_k (k-th largest s_hi) -> band survivors -> exact scores
// (16 lanes per survivor, both planes) -> the k best by (exact score desc, id asc). Raises *overflow (and returns;
// the exact fallback pass then rewrites D/I) when the union or the band does not fit.
template <bool BF>
__global__ void __launch_bounds__(256)
merge_screenk_kernel(const u64* __restrict__ cand, const int* __restrict__ cand_cnt, int G, int k, const float* __restrict__ qbound,
                     const char* __restrict__ Xhi, const char* __restrict__ Xlo, int nkb, const float* __restrict__ q, float* __restrict__ D,
                     long long* __restrict__ I, long long id_offset, int* __restrict__ overflow, int qcap /* queries per group: list stride */, float xs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u64* keys = (u64*)lds;  // [kMergeKLds]
    __shared__ u64 surv[kSurvMax];
    __shared__ int red[4];
    __shared__ int s_n;
    const int ql = blockIdx.x;
    const int tid = threadIdx.x;
    float* Dq = D + (size_t)ql * k;
    long long* Iq = I + (size_t)ql * k;
    const float band2 = 2.f * qbound[ql];

    int total = 0;
    for (int w = tid; w < G; w += 256) total += cand_cnt[(size_t)w * qcap + ql];
    total = block_sum_256(total, red);
    if (tid == 0) { s_n = 0; atomicAdd(overflow + 1, total); }  // telemetry: candidates the main pass handed over (sctl[1])
    __syncthreads();
    const int kk = total < k ? total : k;
    if (kk == 0) {
        for (int i = tid; i < k; i += 256) { Dq[i] = -FLT_MAX; Iq[i] = -1; }
        return;
    }
    if (total > kMergeKLds) {
        if (tid == 0) *overflow = 1;
        return;
    }
    for (int w = 0; w < G; ++w) {
        const int c = cand_cnt[(size_t)w * qcap + ql];
        const u64* lst = cand + ((size_t)w * qcap + ql) * kScreenKCap;
        for (int i = tid; i < c; i += 256) keys[atomicAdd(&s_n, 1)] = lst[i];
    }
    __syncthreads();
    const int S = s_n;  // == total
    u64 hk = 0ull;      // kk-th largest s_hi key of the union
    for (int bit = 63; bit >= 0; --bit) {
        const u64 c = hk | (1ull << bit);
        int n = 0;
        for (int i = tid; i < S; i += 256) n += keys[i] >= c;
        n = block_sum_256(n, red);
        if (n >= kk) hk = c;
    }
    const u64 cut = total < k ? 0ull : floor_key(key_score(hk) - band2);
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = tid; i < S; i += 256)
        if (keys[i] >= cut) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < kSurvMax) surv[pos] = keys[i];
        }
    __syncthreads();
    const int ns = s_n;
    if (ns > kSurvMax) {
        if (tid == 0) *overflow = 1;
        return;
    }
    const int sub = tid & 15;
    const float* qrow = q + (size_t)ql * (nkb * 32);
    for (int c = tid >> 4; c < ns; c += 16) {
        const unsigned row = key_row(surv[c]);
        const float acc = exact_dot16<BF>(Xhi, Xlo, nkb, qrow, row, sub, xs);
        if (sub == 0) surv[c] = make_key(acc, row);  // only this 16-lane group touches surv[c]
    }
    __syncthreads();
    for (int i = tid; i < ns; i += 256) {
        const u64 me = surv[i];
        int rank = 0;
        for (int j = 0; j < ns; ++j) rank += surv[j] > me;
        if (rank < kk) {
            Dq[rank] = key_score(me);
            Iq[rank] = id_offset + (long long)key_row(me);
        }
    }
    for (int i = kk + tid; i < k; i += 256) { Dq[i] = -FLT_MAX; Iq[i] = -1; }
}

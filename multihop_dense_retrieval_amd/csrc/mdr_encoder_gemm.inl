// csrc/mdr_encoder_gemm.inl -- the fp16 MFMA GEMMs: one-tile-per-block kernel (small / medium M), persistent 256x128 kernel, persistent 256x256 kernel, erf-GELU epilogue.
// (The four-wave 256x256 kernel with the generated K-loop lives in mdr_encoder_gemm_quad.inl and shares this file's epilogue helpers.)
// Included by mdr_encoder.hip inside namespace mdr::{anonymous}.
// ---- GEMM: C[M,N] = A[M,K] (fp16, row-major) x W[N,K]^T (fp16, row-major) -----------------------------
// Block tile BM x BN x 64 computed by WGM x WGN waves (wave tile = (BM/WGM) x (BN/WGN) as 16x16 MFMA tiles),
// STAGES-deep LDS ring filled by global_load_lds, one barrier per K-step, counted vmcnt so STAGES-2 stages stay
// in flight across it. Instantiated shapes (launch_gemm picks by problem size):
//   256x256, 4x2 waves (wave 64x128), 2 stages, 128 KiB, 1 block/CU : large M -- halves the L2->LDS bytes per flop
//   128x128, 2x2 waves (wave 64x64),  2 stages,  64 KiB, 2 blocks/CU: medium M
//    64x64,  2x2 waves (wave 32x32),  3 stages,  48 KiB, 3 blocks/CU: small M (hop 1, per-rank slices, CLS projection)
enum { EPI_BIAS_F16 = 0, EPI_BIAS_GELU_F16 = 1, EPI_BIAS_RES_F32 = 2, EPI_BIAS_F32 = 3 };
constexpr int BK = 64;

// erf-GELU(x) = x Phi(x) with the normal tail written as a power of two: Phi(-a) = 2^-(1 + a P(a)), a = |x|, P a degree-5
// polynomial fitted to -log2(erfc(a / sqrt 2)) / a on [0, 6], weighted by the tail itself (scripts/fit_gelu_tail.py; it
// extrapolates monotonically beyond 6). max |Phi error| 2.1e-7, max |GELU error| 6.9e-7 over [-8, 8] evaluated in fp32
// (the Abramowitz-Stegun 7.1.26 erf used before: 2.1e-7): 7 packed FMAs per PAIR of values + one v_exp_f32 + 4 simple ops
// per value, against ~18 ops + v_rcp_f32 + v_exp_f32 per value. Measured (round 2, MDR_GEMM_ABL=5 timeline, FFN1 shape): the
// epilogue of a 256x256 tile 16.4 k -> 14.0 k cycles, the kernel -3 % wall -- the GELU arithmetic was NOT what makes that
// epilogue long (with GELU or without the kernel now takes the same time). libm's erff: ~60 divergent instructions per value.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// d = a * b + (c, c): hipcc scalarises a 2-vector FMA whose addend is a literal (VOP3P takes no literal), so the packed
// form is spelled out with the constant pair in SGPRs
__device__ inline f32x2 pk_fma_c(f32x2 a, f32x2 b, float c) {
    f32x2 d;
    const f32x2 cc = {c, c};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(cc));
    return d;
}
__device__ inline f32x2 gelu_erf2(f32x2 x) {
    const f32x2 a = __builtin_elementwise_abs(x);
    f32x2 p = pk_fma_c(a, (f32x2){-1.982813420e-05f, -1.982813420e-05f}, 6.620948925e-04f);
    p = pk_fma_c(p, a, -7.759194708e-03f);
    p = pk_fma_c(p, a, 5.296392132e-02f);
    p = pk_fma_c(p, a, 4.590664427e-01f);
    p = pk_fma_c(p, a, 1.151119066e+00f);
    const f32x2 e = pk_fma_c(p, a, 1.0f);
    f32x2 t;
    t[0] = __builtin_amdgcn_exp2f(-e[0]);  // Phi(-|x|); raw v_exp_f32: the argument is <= -1, underflow to 0 is the right answer
    t[1] = __builtin_amdgcn_exp2f(-e[1]);
    const f32x2 s = __builtin_elementwise_copysign(0.5f - t, x);  // Phi(x) - 1/2
    return x * s + 0.5f * x;
}
__device__ inline f32x4 gelu_erf4(f32x4 x) {
    const f32x2 lo = gelu_erf2((f32x2){x[0], x[1]}), hi = gelu_erf2((f32x2){x[2], x[3]});
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// the epilogues' LDS scratch is written as halves or floats and read back as raw 16-byte chunks by OTHER lanes: accesses through these types may
// alias anything, so the compiler keeps a block's reads in front of the next block's writes (with plain vector types it is free to swap them)
typedef u32x2 __attribute__((may_alias)) scr_u32x2;
typedef u32x4 __attribute__((may_alias)) scr_u32x4;
typedef f32x4 __attribute__((may_alias)) scr_f32x4;
// four floats -> four halves (round to nearest even, what (_Float16)x does) as two v_cvt_pk_f16_f32. Through __builtin_convertvector, NOT inline asm:
// gfx950 needs a wait state between a VALU write and a v_cvt_pk_f16_f32 that reads it, which hipcc inserts for its own instructions only (an asm
// version returned wrong halves in ~0.05 % of the outputs).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ inline u32x2 cvt_pk_half4(f32x4 v) {
    const half2v lo = __builtin_convertvector((f32x2){v[0], v[1]}, half2v), hi = __builtin_convertvector((f32x2){v[2], v[3]}, half2v);
    return (u32x2){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}

template <int BM_, int BN_, int WGM_, int WGN_, int STAGES_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, STAGES = STAGES_;
    static constexpr int THREADS = 64 * WGM * WGN;
    static constexpr int MT = BM / WGM / 16, NT = BN / WGN / 16;  // 16x16 tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_CHUNKS = BM * 8 / THREADS, W_CHUNKS = BN * 8 / THREADS;  // 16-B DMA pieces per thread per stage
    static constexpr int PER_STAGE = A_CHUNKS + W_CHUNKS;
    static_assert(BM * 8 % THREADS == 0 && BN * 8 % THREADS == 0, "tile must split evenly over the threads");
};

template <int EPI, typename C>
__global__ void __launch_bounds__(C::THREADS)
gemm_f16_kernel(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M_cap,
                const int* __restrict__ M_dev, int N, int K, void* __restrict__ out, int ldo, const _Float16* __restrict__ res, int ldr) {
    constexpr int STAGES = C::STAGES, MT = C::MT, NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int M = M_dev ? min(*M_dev, M_cap) : M_cap;
    const int ntn = N / C::BN;
    const int bid = blockIdx.x;  // n fastest: the blocks sharing an A tile are launched together (measured best)
    const int m0 = (bid / ntn) * C::BM, n0 = (bid % ntn) * C::BN;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int g = lane >> 4, lr = lane & 15;

    // DMA plan: LDS slot p (16 B) of a tile holds global chunk (row = p>>3, k-slot = (p&7) ^ (row&7)): the LDS image is
    // linear in lane order (what global_load_lds writes), the XOR swizzle lives in the SOURCE address (guide rule 21)
    const _Float16* a_src[C::A_CHUNKS];
    const _Float16* w_src[C::W_CHUNKS];
#pragma unroll
    for (int i = 0; i < C::A_CHUNKS; ++i) {
        const int p = i * C::THREADS + tid;
        const int row = p >> 3, s = (p & 7) ^ (row & 7);
        int ar = m0 + row;
        ar = ar < M ? ar : M - 1;  // rows past M are computed on a valid row and never stored
        a_src[i] = A + (size_t)ar * lda + s * 8;
    }
#pragma unroll
    for (int i = 0; i < C::W_CHUNKS; ++i) {
        const int p = i * C::THREADS + tid;
        const int row = p >> 3, s = (p & 7) ^ (row & 7);
        w_src[i] = W + (size_t)(n0 + row) * K + s * 8;
    }
    auto issue = [&](int stage, int k0) {
        char* base = lds + stage * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::A_CHUNKS; ++i)
            __builtin_amdgcn_global_load_lds(MDR_GPTR(a_src[i] + k0), MDR_LPTR(base + (i * C::THREADS + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::W_CHUNKS; ++i)
            __builtin_amdgcn_global_load_lds(MDR_GPTR(w_src[i] + k0), MDR_LPTR(base + C::A_BYTES + (i * C::THREADS + wave * 64) * 16), 16, 0, 0);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int sw0 = ((0 * 4 + g) ^ (lane & 7)) << 4, sw1 = ((1 * 4 + g) ^ (lane & 7)) << 4;
    const int a_off = (wm * MT * 16 + lr) * 128, w_off = C::A_BYTES + (wn * NT * 16 + lr) * 128;

    const int KT = K / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) issue(s, s * BK);
    for (int kt = 0; kt < KT; ++kt) {
        // stage kt has landed; at most STAGES-2 younger stages stay in flight across the barrier
        const int younger = min(STAGES - 2, KT - 1 - kt);
        if (STAGES >= 4 && younger >= 2)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::PER_STAGE * 2 < 64 ? C::PER_STAGE * 2 : 63) : "memory");
        else if (STAGES >= 3 && younger >= 1)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::PER_STAGE) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + STAGES - 1 < KT) issue((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * BK);
        const char* base = lds + (kt % STAGES) * C::STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sw = s ? sw1 : sw0;
            half8 af[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = *(const half8*)(base + a_off + t * 16 * 128 + sw);
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const half8*)(base + w_off + t * 16 * 128 + sw);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    }

    // epilogue: lane holds C[m = .. + lr][n = .. + 4g + r], r = 0..3 (first MFMA operand = W rows)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + (wm * MT + mt) * 16 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + (wn * NT + nt) * 16 + 4 * g;
            const f32x4 b4 = *(const f32x4*)(bias + n);
            f32x4 v = acc[mt][nt] + b4;
            if (EPI == EPI_BIAS_GELU_F16) {
                v = gelu_erf4(v);
            }
            if (EPI == EPI_BIAS_RES_F32) {
                const half4 r4 = *(const half4*)(res + (size_t)m * ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
            }
            if (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) {
                half4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                *(half4*)((_Float16*)out + (size_t)m * ldo + n) = o;
            } else {
                *(f32x4*)((float*)out + (size_t)m * ldo + n) = v;
            }
        }
    }
}

// ---- tail of the persistent 256x256 kernels: the last, partial round on 128x128 tiles ------------------------------------------
// T = ntm x ntn tiles of 256x256 over G persistent workgroups take ceil(T / G) rounds; when T is a little over a multiple of G the last round
// keeps a few workgroups busy for a whole tile time while the rest of the chip idles (86 row tiles of 256 x N = 768: 258 tiles = TWO rounds on 256
// CUs for one round of work -- the +37 % step at 21.8 k tokens of DESIGN.md section 4). gemm_head_row_tiles() gives the 256x256 walk only the row
// tiles that fill COMPLETE rounds; the remaining rows (fewer than max_rem + ntn tiles' worth) are computed by the same workgroups afterwards on
// 128x128 tiles (gemm_tail_tile: the one-tile-per-block kernel's K-loop on a four-slot ring over the whole 128 KiB), which spreads them over 4x as many units. Same K order and MFMA per output element as every other flavour: bit-identical results.
__host__ __device__ inline int gemm_head_row_tiles(int ntm, int ntn, int G, int max_rem) {
    const long long T = (long long)ntm * ntn;
    const long long full = T / G, rem = T % G;
    if (full < 1 || rem == 0 || rem > max_rem) return ntm;
    return (int)(full * G / ntn);
}

// The tail's MFMAs are spelled as asm with VGPR accumulators: gemm_quad_kernel keeps a[0:255] live ACROSS its asm statements, so the compiler must not
// be given a reason to touch an AGPR anywhere in that kernel (its own MFMA form put these 64 accumulators there: scripts/check_quad_agprs.py). Hazards
// the compiler would otherwise cover: the accumulate chain has SrcC == vDst of the same opcode (no wait states needed); the first read of an accumulator
// by the VALU comes behind gemm_tail_settle() (32 wait states > the 8 passes of the last MFMA); operands come from LDS (the compiler waits on lgkmcnt
// for asm inputs as for any other use).
__device__ __forceinline__ void mfma16_vgpr(f32x4& acc, half8 a, half8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void gemm_tail_settle(f32x4 (&acc)[4][4]) {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                   "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
}

// One 128x128 tile by WGM x WGN waves (tg = thread 0..64 WGM WGN - 1, wg = wave index) on a ring of SLOTS 32 KiB stages at `ring` (SLOTS - 1 DMA batches in
// flight: with two slots a K-step is one L2 round trip, 1.3 us measured; four bring it to the LDS / MFMA time). Contains workgroup barriers: every wave of the
// WORKGROUP must call it the same number of times with the same K. A call without a tile (m0 >= M): loads on the clamped last row, no stores.
template <int EPI, int WGM, int WGN, int SLOTS>
__device__ __forceinline__ void gemm_tail_tile(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M,
                                               int K, void* __restrict__ out, int ldo, int m0, int n0, char* ring, int tg, int wg, int lane) {
    constexpr int BM = 128, BN = 128, THREADS = 64 * WGM * WGN, MT = BM / WGM / 16, NT = BN / WGN / 16, A_CHUNKS = BM * 8 / THREADS, W_CHUNKS = BN * 8 / THREADS;
    constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2, PER_STAGE = A_CHUNKS + W_CHUNKS;
    static_assert(MT * NT <= 16 && SLOTS >= 2 && SLOTS <= 4, "accumulators are settled sixteen at a time; the counted waits are written for 2-4 slots");
    const int wm = wg / WGN, wn = wg % WGN;
    const int g = lane >> 4, lr = lane & 15;
    const _Float16* a_src[A_CHUNKS];
    const _Float16* w_src[W_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        const int p = i * THREADS + tg;
        const int row = p >> 3, s = (p & 7) ^ (row & 7);
        int ar = m0 + row;
        ar = ar < M ? ar : M - 1;
        a_src[i] = A + (size_t)ar * lda + s * 8;
    }
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) {
        const int p = i * THREADS + tg;
        const int row = p >> 3, s = (p & 7) ^ (row & 7);
        w_src[i] = W + (size_t)(n0 + row) * K + s * 8;
    }
    auto issue = [&](int stage, int k0) __attribute__((always_inline)) {
        char* base = ring + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
            __builtin_amdgcn_global_load_lds(MDR_GPTR(a_src[i] + k0), MDR_LPTR(base + (i * THREADS + wg * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i)
            __builtin_amdgcn_global_load_lds(MDR_GPTR(w_src[i] + k0), MDR_LPTR(base + A_BYTES + (i * THREADS + wg * 64) * 16), 16, 0, 0);
    };
    f32x4 acc[4][4];  // [MT][NT] used; the settle helper takes all sixteen
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_tail_settle(acc);  // (the zeros are VALU writes: keep them well ahead of the first asm MFMA that reads them as SrcC)
    const int sw0 = ((0 * 4 + g) ^ (lane & 7)) << 4, sw1 = ((1 * 4 + g) ^ (lane & 7)) << 4;
    const int a_off = (wm * MT * 16 + lr) * 128, w_off = A_BYTES + (wn * NT * 16 + lr) * 128;
    const int KT = K / BK;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody is done with the ring (the previous tile's last K-step, or the 256x256 walk)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s_ = 0; s_ < SLOTS - 1; ++s_)
        if (s_ < KT) issue(s_, s_ * BK);
    for (int kt = 0; kt < KT; ++kt) {
        // stage kt has landed; at most SLOTS - 2 younger stages stay in flight across the barrier
        const int younger = min(SLOTS - 2, KT - 1 - kt);
        if (SLOTS >= 4 && younger >= 2)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_STAGE * 2) : "memory");
        else if (SLOTS >= 3 && younger >= 1)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_STAGE) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + SLOTS - 1 < KT) issue((kt + SLOTS - 1) % SLOTS, (kt + SLOTS - 1) * BK);
        const char* base = ring + (kt % SLOTS) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sw = s ? sw1 : sw0;
            half8 af[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = *(const half8*)(base + a_off + t * 16 * 128 + sw);
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const half8*)(base + w_off + t * 16 * 128 + sw);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mfma16_vgpr(acc[mt][nt], wf[nt], af[mt]);
        }
    }
    gemm_tail_settle(acc);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + (wm * MT + mt) * 16 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + (wn * NT + nt) * 16 + 4 * g;
            const f32x4 b4 = *(const f32x4*)(bias + n);
            f32x4 v = acc[mt][nt] + b4;
            if (EPI == EPI_BIAS_GELU_F16) v = gelu_erf4(v);
            if (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) {
                *(scr_u32x2*)((_Float16*)out + (size_t)m * ldo + n) = cvt_pk_half4(v);
            } else {
                *(f32x4*)((float*)out + (size_t)m * ldo + n) = v;
            }
        }
    }
}

// ---- persistent GEMM for large M ------------------------------------------------------------------------
// K is short here (768 or 3072): a one-tile-per-block kernel spends as long filling and draining its LDS ring
// as computing. This kernel keeps ONE 512-thread block per CU alive and walks (tile, k-step) as one flat stream:
// the loads of the next tile's first stages are issued during the current tile's last K-steps, so the
// global_load_lds pipeline never empties; the epilogue (bias from LDS, no register-destination VMEM load that
// would make the compiler wait on the in-flight DMA) runs under the next tile's loads.
// Tile 256x128x64, 8 waves as 4x2 (wave 64x64), 3-slot ring (2 batches in flight), bias vector staged in LDS once.
// Measured alternatives (round 1, hop-2 shapes): 256x256 with a 2-slot ring 8.3 ms vs 5.8 ms (one batch in flight is not
// enough), 256x256 with four K=32 slots 7.4 ms (twice the barriers), burst-issued DMA +2 %, n-fastest tile order +9 %.
// The residual of EPI_BIAS_RES_F32 is NOT added here: the LayerNorm kernel that follows adds it (res argument).
using GemmP = GemmCfg<256, 128, 4, 2, 3>;   // 3 slots of 48 KiB: two batches in flight
#ifndef MDR_GEMM_EPI
#define MDR_GEMM_EPI 2
#endif
constexpr int kPersistBiasMax = 3072;  // floats of bias kept in LDS behind the ring (12 KiB)

template <int EPI, typename C>
__global__ void __launch_bounds__(C::THREADS)
gemm_persist_kernel(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M_cap,
                    const int* __restrict__ M_dev, int N, int K, void* __restrict__ out, int ldo, int epi_mode) {
    static_assert(C::MT == 4 && (C::NT == 4 || C::NT == 8) && C::A_CHUNKS == 4 && C::W_CHUNKS <= 4, "the pinned K-step below is written for these shapes");
    constexpr int MT = C::MT, NT = C::NT, SLOTS = C::STAGES, AHEAD = SLOTS - 1;  // batches issued ahead of the one being computed
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = (float*)(lds + C::LDS_BYTES);
    const int M = M_dev ? min(*M_dev, M_cap) : M_cap;
    // XCD-aware tile assignment. Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only) and each XCD
    // has a private 4 MiB L2. Tiles are numbered m-major (n fastest); XCD x owns the contiguous eighth [T x / 8, T (x+1) / 8)
    // of that order and its G / 8 workgroups walk it round-robin: the workgroups sharing an A tile run on ONE L2 at the same
    // time (without that every XCD pulled all of A through the fabric: measured 8x the algorithmic A traffic), and no XCD
    // has more than one tile above the average -- the earlier gm x gn grid of XCDs lost up to a whole round of tiles to
    // rounding at 20 k rows (3 rounds instead of 2 for the out-projection).
    const int ntn = N / C::BN, ntm = (M + C::BM - 1) / C::BM;
    const long long T_all = (long long)ntm * ntn;
    const int xcd = blockIdx.x & 7;
    const int t_base = (int)(T_all * xcd / 8), local_tiles = (int)(T_all * (xcd + 1) / 8) - t_base;
    const int lb = blockIdx.x >> 3, G = gridDim.x >> 3;  // this XCD's workgroups
    if (lb >= local_tiles) return;
    const int n_my = (local_tiles - lb + G - 1) / G;
    auto tile_origin = [&](int j, int& m0, int& n0) {
        const int t = t_base + lb + j * G;
        m0 = (t / ntn) * C::BM;
        n0 = (t % ntn) * C::BN;
    };
    const int KT = K / BK;
    const int total_steps = n_my * KT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int g = lane >> 4, lr = lane & 15;

    for (int i = tid; i < N; i += C::THREADS) lds_bias[i] = bias[i];  // retired by the first barrier wait below

    // ---- loader state: (tile, k-step) the next DMA batch belongs to ----
    const _Float16* a_src[C::A_CHUNKS];
    const _Float16* w_src[C::W_CHUNKS];
    int ld_tile = 0, ld_kt = 0, ld_step = 0;
    auto set_load_tile = [&](int j) {
        int m0, n0;
        tile_origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < C::A_CHUNKS; ++i) {
            const int p = i * C::THREADS + tid;
            const int row = p >> 3, s = (p & 7) ^ (row & 7);
            int ar = m0 + row;
            ar = ar < M ? ar : M - 1;
            a_src[i] = A + (size_t)ar * lda + s * 8;
        }
#pragma unroll
        for (int i = 0; i < C::W_CHUNKS; ++i) {
            const int p = i * C::THREADS + tid;
            const int row = p >> 3, s = (p & 7) ^ (row & 7);
            w_src[i] = W + (size_t)(n0 + row) * K + s * 8;
        }
    };
    // advance the loader to its next batch (branchy part, kept OUT of the pinned MFMA block below). Past the end of
    // the stream the pointers simply stay on the last tile: the surplus batches land in free slots and are never read,
    // which keeps the DMA count per K-step constant (vmcnt(PER_STAGE) is then exact at every step).
    auto advance_loader = [&]() {
        if (ld_step < total_steps && ld_kt == 0) set_load_tile(ld_tile);
    };
    auto loader_done = [&]() {
        ++ld_step;
        if (ld_step < total_steps && ++ld_kt == KT) { ld_kt = 0; ++ld_tile; }
    };

    const int sw0 = ((0 * 4 + g) ^ (lane & 7)) << 4, sw1 = ((1 * 4 + g) ^ (lane & 7)) << 4;
    const int a_off = (wm * MT * 16 + lr) * 128, w_off = C::A_BYTES + (wn * NT * 16 + lr) * 128;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the bias loads above, before any DMA is in flight
#pragma unroll
    for (int pre = 0; pre < AHEAD; ++pre) {
        advance_loader();
        char* base = lds + (ld_step % SLOTS) * C::STAGE_BYTES;
        const int k0 = ld_kt * BK;
#pragma unroll
        for (int i = 0; i < C::A_CHUNKS; ++i)
            __builtin_amdgcn_global_load_lds(MDR_GPTR(a_src[i] + k0), MDR_LPTR(base + (i * C::THREADS + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::W_CHUNKS; ++i)
            __builtin_amdgcn_global_load_lds(MDR_GPTR(w_src[i] + k0), MDR_LPTR(base + C::A_BYTES + (i * C::THREADS + wave * 64) * 16), 16, 0, 0);
        loader_done();
    }
    // Deferred epilogue: the finished tile's results wait in registers (bias / GELU applied, converted to the output type)
    // and leave two fragments per K-step at the TOP of the next tile's steps, before that step's DMA pieces. Stored right
    // after the tile instead, the 16 stores are the youngest VMEM ops at the next counted vmcnt wait, which then has to
    // drain every DMA batch in flight plus the stores (measured: 25-30 % of the K = 768 GEMMs).
    constexpr bool F16OUT = EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16;
    using pend_t = typename std::conditional<F16OUT, half4, f32x4>::type;
    pend_t pend[MT * NT];
    int pend_m0 = 0, pend_n0 = 0, pend_left = 0;  // fragments of the previous tile not stored yet (wave-uniform)
    bool pend_full = false;      // that tile lies completely below M: its stores are unconditional, so their COUNT is known
    bool stored_two = false;     // the previous K-step issued exactly two (unconditional) stores before its DMA pieces
    auto store_pending = [&](int f, bool check) __attribute__((always_inline)) {  // f compile-time after inlining
        const int mt = f / NT, nt = f % NT;
        const int m = pend_m0 + (wm * MT + mt) * 16 + lr;
        const int n = pend_n0 + (wn * NT + nt) * 16 + 4 * g;
        if (!check || m < M) {
            if (F16OUT) *(pend_t*)((_Float16*)out + (size_t)m * ldo + n) = pend[f];
            else *(pend_t*)((float*)out + (size_t)m * ldo + n) = pend[f];
        }
    };
    auto flush_pending = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < MT * NT; ++f)
            if (f >= MT * NT - pend_left) store_pending(f, true);
        pend_left = 0;
    };
    int step = 0;
    for (int j = 0; j < n_my; ++j) {
        int m0, n0;
        tile_origin(j, m0, n0);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt, ++step) {
            // batch `step` has landed; exactly one younger batch stays in flight across the barrier. Epilogue stores
            // issued since only make this wait more conservative (vmcnt completes in order).
            // (VMEM ops complete in order: when the previous step put exactly two stores in front of its DMA pieces they may
            // stay outstanding with them -- the stores then have two K-steps to be acknowledged instead of one.)
            if (stored_two)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::PER_STAGE * (AHEAD - 1) + 2) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::PER_STAGE * (AHEAD - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stored_two = false;
            if (pend_left > 0) {  // wave-uniform; fragments leave in order 0, 1, 2, ..: two per step
                const bool chk = !pend_full;
                stored_two = pend_full && epi_mode == 2;
                switch (MT * NT - pend_left) {
#define MDR_PEND_CASE(F) case F: if (F + 1 < MT * NT) { \
        if (chk) { store_pending(F + 1 < MT * NT ? F : 0, true); store_pending(F + 1 < MT * NT ? F + 1 : 0, true); } \
        else { store_pending(F + 1 < MT * NT ? F : 0, false); store_pending(F + 1 < MT * NT ? F + 1 : 0, false); } } break;
                    MDR_PEND_CASE(0) MDR_PEND_CASE(2) MDR_PEND_CASE(4) MDR_PEND_CASE(6) MDR_PEND_CASE(8) MDR_PEND_CASE(10) MDR_PEND_CASE(12) MDR_PEND_CASE(14)
                    MDR_PEND_CASE(16) MDR_PEND_CASE(18) MDR_PEND_CASE(20) MDR_PEND_CASE(22) MDR_PEND_CASE(24) MDR_PEND_CASE(26) MDR_PEND_CASE(28) MDR_PEND_CASE(30)
#undef MDR_PEND_CASE
                    default: break;
                }
                pend_left -= 2;
            }
            advance_loader();
            // ---- one straight-line block: LDS fragment reads, 32 MFMAs, and the DMA pieces of batch step+AHEAD (into the
            // slot read at step-1, free since the barrier) spread BETWEEN the MFMAs. Issued in a burst right after the
            // barrier they cost each wave ~1k cycles of VMEM issue stall while both waves of a SIMD sit idle.
            const char* base = lds + (step % SLOTS) * C::STAGE_BYTES;
            char* lbase = lds + (ld_step % SLOTS) * C::STAGE_BYTES;
            const int k0 = ld_kt * BK;
            half8 af0[MT], wf0[NT], af1[MT], wf1[NT];
#pragma unroll
            for (int q = 0; q < MT; ++q) af0[q] = *(const half8*)(base + a_off + q * 16 * 128 + sw0);
#pragma unroll
            for (int q = 0; q < NT; ++q) wf0[q] = *(const half8*)(base + w_off + q * 16 * 128 + sw0);
            __builtin_amdgcn_sched_barrier(0);
            // {4 MFMA (k-sub 0), 2 fragment reads for k-sub 1, 1 DMA piece} x 4, order pinned with hard scheduling barriers
            // (sched_group_barrier does not move global_load_lds: it stays a burst)
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[grp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[nt], af0[grp], acc[grp][nt], 0, 0, 0);
                af1[grp] = *(const half8*)(base + a_off + grp * 16 * 128 + sw1);
                wf1[grp] = *(const half8*)(base + w_off + grp * 16 * 128 + sw1);
                if (NT == 8) wf1[4 + grp] = *(const half8*)(base + w_off + (4 + grp) * 16 * 128 + sw1);
                __builtin_amdgcn_global_load_lds(MDR_GPTR(a_src[grp] + k0), MDR_LPTR(lbase + (grp * C::THREADS + wave * 64) * 16), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[grp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[nt], af1[grp], acc[grp][nt], 0, 0, 0);
                if (grp < C::W_CHUNKS)
                    __builtin_amdgcn_global_load_lds(MDR_GPTR(w_src[grp] + k0), MDR_LPTR(lbase + C::A_BYTES + (grp * C::THREADS + wave * 64) * 16), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            loader_done();
        }
        // epilogue values -> pending registers (lane holds C[m = .. + lr][n = .. + 4g + r]); stores are deferred, see above
        if (pend_left > 0) flush_pending();  // K shorter than 8 steps: the previous tile still has fragments left
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + (wn * NT + nt) * 16 + 4 * g;
                const f32x4 b4 = *(const f32x4*)(lds_bias + n);
                f32x4 v = acc[mt][nt] + b4;
                if (EPI == EPI_BIAS_GELU_F16) {
                    v = gelu_erf4(v);
                }
                if constexpr (F16OUT) {
                    half4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                    pend[mt * NT + nt] = o;
                } else {
                    pend[mt * NT + nt] = v;
                }
            }
        }
        pend_m0 = m0;
        pend_n0 = n0;
        pend_left = MT * NT;
        pend_full = m0 + C::BM <= M;
        if (epi_mode == 0) flush_pending();  // measurement: store right after the tile
    }
    flush_pending();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus batches must have landed before the LDS is released
}


// ---- persistent 256x256x64 GEMM, one phase for all waves, half-step refill ------------------------------------
// gemm_persist_kernel's structure (every wave interleaves its own DMA pieces with its MFMAs, so the L2->LDS path is
// always fed) on a 256x256 tile, which moves 1/3 fewer L2->LDS bytes per flop -- the measured bound of that kernel.
// Only two 64 KiB slots fit in LDS; what makes two enough is that a K-step consumes its slot early: the step is four
// sub-phases of 16 MFMAs per wave, (k-half 0 | 1) x (m-fragments 0-3 | 4-7), and the LDS reads of sub-phase i+1 are
// issued at the top of sub-phase i, so after barrier B (top of sub-phase 4) the slot is free and its refill with
// K-tile T+2 starts while the MFMAs of step T are still running:
//   barrier A (top of step T, after vmcnt(2)): K-tile T landed            barrier B: every wave's reads of slot s retired
//   DMA pieces (8 per wave and step, 2 per sub-phase): sub-phases 1-3 of step T carry K-tile T+1 (slot s^1, freed at
//   barrier B of step T-1), sub-phase 4 the first quarter of K-tile T+2 (slot s).
// Measured (round 1, scripts/measure/gpu_gemm_bench.py): 9-31 % faster than gemm_persist_kernel at 65k rows (qkv 286 vs 350 us,
// ffn1 382 vs 499 us); at 20k rows the 256x256 tile count quantises badly over 8 XCDs x 32 workgroups, so launch_gemm
// compares the two kernels' round counts per call. A ping-pong variant (two wave groups half a phase apart, 8 barrier
// intervals of 16 MFMAs per K-tile) was slower than both: a barrier interval cost 650-850 cycles against the 256 of
// its MFMAs whichever of DMA / MFMA / LDS reads was removed -- the barrier skeleton itself; removed.
using GemmB2 = GemmCfg<256, 256, 2, 4, 2>;

// ABL = the COMPILE-TIME macro MDR_GEMM_ABL of a measurement build (`build.py -DMDR_GEMM_ABL=n --out=libmdrhip_abl.so`, selected with
// MDR_LIB_PATH by scripts/measure/gpu_gemm_bench.py); the product library is built with 0 and holds none of this. Results are wrong for
// ABL != 0 except 5: 1 = no DMA after the prologue, 2 = no LDS fragment reads, 3 = no MFMAs, 4 = no epilogue stores -- which of the
// CU's pipes the K-loop is waiting for. 5 = correct results + an s_memtime timeline of wave 0 of every workgroup summed into
// g_gemm_stamp (mdr_test_gemm_stamps, include/mdr_hip_measure.h): [0] wait + barrier A, [1] sub-phase 1 (incl. its fragment reads),
// [2] sub-phase 2, [3] sub-phase 3, [4] wait + barrier B, [5] sub-phase 4, [6] epilogue, [7] K-tiles counted.
#ifndef MDR_GEMM_ABL
#define MDR_GEMM_ABL 0
#endif
#if MDR_GEMM_ABL == 5
__device__ unsigned long long g_gemm_stamp[8];
#endif
template <int EPI, int ABL = MDR_GEMM_ABL>
__global__ void __launch_bounds__(512)
gemm_big_kernel(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M_cap,
                const int* __restrict__ M_dev, int N, int K, void* __restrict__ out, int ldo) {
    using C = GemmB2;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = (float*)(lds + C::LDS_BYTES);
    const int M = M_dev ? min(*M_dev, M_cap) : M_cap;
    // output descriptor: raw buffer over rows [0, M) -- stores to rows past M (the last, partial row tile) are dropped by its bounds check
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(((unsigned)(M - 1) * (unsigned)ldo + (unsigned)N) * ((EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) ? 2u : 4u)), 0x00020000);
    const int ntn = N / 256;
    const int ntm = gemm_head_row_tiles((M + 255) / 256, ntn, gridDim.x, gridDim.x / 2);  // row tiles of the 256x256 walk; the rows behind them: gemm_tail_tile, below
    const long long T_all = (long long)ntm * ntn;  // tile order and XCD ownership: see gemm_persist_kernel
    const int xcd = blockIdx.x & 7;
    const int t_base = (int)(T_all * xcd / 8), local_tiles = (int)(T_all * (xcd + 1) / 8) - t_base;
    const int lb = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int tail_m0 = ntm * 256;
    if (lb >= local_tiles && tail_m0 >= M) return;
    const int n_my = lb < local_tiles ? (local_tiles - lb + G - 1) / G : 0;
    auto tile_origin = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = t_base + lb + j * G;
        m0 = (t / ntn) * 256;
        n0 = (t % ntn) * 256;
    };
    const int KT = K / BK;
    const int total = n_my * KT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // 2 (m) x 4 (n) waves, 128 x 64 outputs each
    const int g = lane >> 4, lr = lane & 15;

    for (int i = tid; i < N; i += 512) lds_bias[i] = bias[i];

    // ---- loader: piece i of A / W = rows 64 i + (tid >> 3), 16-B chunk (tid & 7) ^ (row & 7) ----
    const int ld_row = tid >> 3;
    const int ld_chunk = ((tid & 7) ^ (ld_row & 7)) * 8;
    unsigned a_off[4];  // element offsets (rows clamped to M - 1)
    unsigned w_off = 0;
    int ld_tile = 0, ld_kt = 0, ld_T = 0;  // K-tile the NEXT quarter (2 pieces per wave-thread) belongs to
    auto set_load_tile = [&](int j) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ar = m0 + 64 * i + ld_row;
            ar = ar < M ? ar : M - 1;
            a_off[i] = (unsigned)ar * (unsigned)lda + (unsigned)ld_chunk;
        }
        w_off = (unsigned)(n0 + ld_row) * (unsigned)K + (unsigned)ld_chunk;
    };
    // piece c (0..7) of the loader's K-tile: 0-3 = A rows 64c.., 4-7 = W rows 64(c-4)..
    bool dma_on = true;
    auto issue_piece = [&](int c) __attribute__((always_inline)) {
        if (ABL == 1 && !dma_on) return;
        char* slot = lds + (ld_T & 1) * C::STAGE_BYTES;
        const int k0 = ld_kt * BK;
        const _Float16* src = c < 4 ? A + (a_off[c] + (unsigned)k0) : W + (w_off + (unsigned)(64 * (c - 4) * K + k0));
        char* dst = slot + (c < 4 ? 0 : C::A_BYTES) + ((c & 3) * 512 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds(MDR_GPTR(src), MDR_LPTR(dst), 16, 0, 0);
    };
    auto next_ktile = [&]() __attribute__((always_inline)) {
        ++ld_T;
        if (ld_T < total) {
            if (++ld_kt == KT) { ld_kt = 0; ++ld_tile; set_load_tile(ld_tile); }
        }
    };

    const int a_rd = (wr * 128 + lr) * 128, w_rd = C::A_BYTES + (wc * 64 + lr) * 128;
    const int sw0 = ((0 * 4 + g) ^ (lane & 7)) << 4, sw1 = ((1 * 4 + g) ^ (lane & 7)) << 4;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bias loads, before any DMA is in flight
    if (n_my > 0) {
    set_load_tile(0);
    // prologue: K-tile 0 completely, then the first 6 pieces of K-tile 1 (what sub-phases 1-3 of a step -1 would have issued)
#pragma unroll
    for (int c = 0; c < 8; ++c) issue_piece(c);
    next_ktile();
#pragma unroll
    for (int c = 0; c < 2; ++c) issue_piece(c);  // "sub-phase 4 of step -1"
    }
    dma_on = false;

    f32x4 acc[8][4];
    half8 wf0[4], wf1[4], af_a[4], af_b[4];
    if (ABL == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wf0[q] = wf1[q] = af_a[q] = af_b[q] = (half8){1, 1, 1, 1, 1, 1, 1, 1};
    }
    // 16 MFMAs (4 m-fragments from mbase x 4 n-fragments) with DMA pieces pc, pc+1 pinned after the 2nd and 4th group
    auto sub_phase = [&](auto zero_c, int mbase, const half8* wfr, const half8* afr, int pc) __attribute__((always_inline)) {
        constexpr bool Z = decltype(zero_c)::value;  // first K-tile of an output tile: accumulate onto 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (ABL == 3) { asm volatile("" ::"v"(wfr[n]), "v"(afr[q])); if (Z) acc[mbase + q][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
                acc[mbase + q][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfr[n], afr[q], Z ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mbase + q][n], 0, 0, 0);
            }
            if (q == 1) issue_piece(pc);
            if (q == 3) issue_piece(pc + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int kt = 0, tile = 0;
    unsigned long long stamp_sum[7] = {0, 0, 0, 0, 0, 0, 0}, stamp_t = 0;
    auto stamp = [&](int seg) __attribute__((always_inline)) {
        if (ABL != 5) return;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_readcyclecounter();
        if (seg >= 0) stamp_sum[seg] += now - stamp_t;
        stamp_t = now;
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int T = 0; T < total; ++T) {
        const char* slot = lds + (T & 1) * C::STAGE_BYTES;
        const bool first = kt == 0;
        stamp(T == 0 ? -1 : 6);
        // barrier A: K-tile T landed (at most the 2 pieces issued in the previous sub-phase 4 may still fly)
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(0);
        // reads for sub-phases 1 and 2
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ABL != 2) wf0[q] = *(const half8*)(slot + w_rd + q * 16 * 128 + sw0);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ABL != 2) af_a[q] = *(const half8*)(slot + a_rd + q * 16 * 128 + sw0);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ABL != 2) af_b[q] = *(const half8*)(slot + a_rd + (4 + q) * 16 * 128 + sw0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 1: k-half 0, m-fragments 0-3; pieces 2,3 of the loader's K-tile (T+1)
        if (first) sub_phase(std::true_type{}, 0, wf0, af_a, 2);
        else sub_phase(std::false_type{}, 0, wf0, af_a, 2);
        stamp(1);
        // reads for sub-phase 3 (k-half 1): W fragments, A fragments 0-3 into the registers sub-phase 1 just released
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ABL != 2) wf1[q] = *(const half8*)(slot + w_rd + q * 16 * 128 + sw1);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ABL != 2) af_a[q] = *(const half8*)(slot + a_rd + q * 16 * 128 + sw1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 2: k-half 0, m-fragments 4-7; pieces 4,5
        if (first) sub_phase(std::true_type{}, 4, wf0, af_b, 4);
        else sub_phase(std::false_type{}, 4, wf0, af_b, 4);
        stamp(2);
        // reads for sub-phase 4: the LAST reads of this slot
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ABL != 2) af_b[q] = *(const half8*)(slot + a_rd + (4 + q) * 16 * 128 + sw1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 3: k-half 1, m-fragments 0-3; pieces 6,7 complete K-tile T+1
        sub_phase(std::false_type{}, 0, wf1, af_a, 6);
        next_ktile();
        stamp(3);
        // barrier B: every wave's reads of this slot have retired -> it may be refilled (K-tile T+2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(4);
        // ---- sub-phase 4: k-half 1, m-fragments 4-7; pieces 0,1 of K-tile T+2
        sub_phase(std::false_type{}, 4, wf1, af_b, 0);
        stamp(5);
        if (++kt == KT) {
            kt = 0;
            int m0, n0;
            tile_origin(tile, m0, n0);
            ++tile;
            // Epilogue under the loads in flight. The MFMA layout gives a lane 4 consecutive n of ONE row (8 B of f16): stored
            // directly, a wave instruction touches 16 rows x 32 B = 16 partial cache lines, and the 256 such instructions of a
            // tile cost the CU ~25 % of a K = 768 GEMM (measured by skipping them; they are line-REQUEST bound, not byte bound:
            // deferring them over the next K-steps did not help). So each 16-row block goes through a 2 KiB per-wave LDS
            // scratch (16-B chunks XOR-swizzled by row & 7) and leaves as 8 rows x 128 B = 8 full lines per instruction.
            // Round 3: the bias of the wave's 64 columns is read ONCE per tile (hipcc re-read it from LDS in front of every fragment and
            // waited for it each time), the halves are packed by v_cvt_pk_f16_f32 (it emitted 7 instructions per 4 values), and the rows
            // leave through a buffer descriptor whose bounds check (on the VGPR offset) drops the rows past M: no per-row predicate, no 64-bit address
            // arithmetic, and with the branches gone the two reads and two stores of a block are issued together.
            char* scr = lds + C::LDS_BYTES + kPersistBiasMax * 4 + wave * 2048;
            const int rd_row = lane >> 3, rd_chunk = lane & 7;
            constexpr bool F16OUT = EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16;
            constexpr unsigned OB = F16OUT ? 2u : 4u;
            f32x4 b4[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) b4[nt] = *(const f32x4*)(lds_bias + n0 + wc * 64 + nt * 16 + 4 * g);
            // (the descriptor's bounds check covers the VGPR offset only, not the scalar one: the ROW goes into the VGPR, the column base into the SGPR)
            const unsigned o_col = __builtin_amdgcn_readfirstlane((unsigned)(n0 + wc * 64) * OB);
            const unsigned o_lane = (unsigned)(m0 + wr * 128 + rd_row) * (unsigned)ldo * OB + (unsigned)rd_chunk * 16u;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int hf = 0; hf < (F16OUT ? 1 : 2); ++hf) {  // f32 rows of 64 columns take two 128-B passes
#pragma unroll
                    for (int q = 0; q < (F16OUT ? 4 : 2); ++q) {
                        const int nt = F16OUT ? q : 2 * hf + q;
                        f32x4 v = acc[mt][nt] + b4[nt];
                        if (EPI == EPI_BIAS_GELU_F16) {
                            v = gelu_erf4(v);
                        }
                        if constexpr (F16OUT) {
                            *(scr_u32x2*)(scr + lr * 128 + (((q * 2 + (g >> 1)) ^ (lr & 7)) << 4) + (g & 1) * 8) = cvt_pk_half4(v);
                        } else {
                            *(scr_f32x4*)(scr + lr * 128 + (((q * 4 + g) ^ (lr & 7)) << 4)) = v;
                        }
                    }
                    // row r = rd_row (+8), 16-B chunk rd_chunk of the 128-B row block
                    u32x4 val[2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int r = rd_row + 8 * half;
                        val[half] = *(const scr_u32x4*)(scr + r * 128 + ((rd_chunk ^ (r & 7)) << 4));
                    }
                    if (ABL == 4) { asm volatile("" ::"v"(val[0]), "v"(val[1])); continue; }
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        __builtin_amdgcn_raw_buffer_store_b128(val[half], out_rsrc, o_lane + (unsigned)(mt * 16 + 8 * half) * (unsigned)ldo * OB, o_col + (unsigned)hf * 128u, 0);
                    // A 16-byte buffer store reads its data registers over several cycles; hipcc's hazard recogniser assumes that a store with an
                    // SGPR offset is exempt and lets the next block's first v_pk_add overwrite them straight away -- on gfx950 that corrupted the last
                    // dword of lanes 12-15 of every 16 (found as wrong values in odd rows of the last column of a chunk). Three wait states, with the
                    // data as operands of the asm so that the registers stay untouched until they have passed.
                    asm volatile("s_nop 2" ::"v"(val[0]), "v"(val[1]) : "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus pieces must have landed before the LDS is released
    if (tail_m0 < M) {  // (uniform over the grid) the partial last round, on 128x128 tiles: all eight waves (4 x 2) on one tile, four 32 KiB slots
        const int ttn = N / 128, Tt = ((M - tail_m0 + 127) / 128) * ttn;
        for (int t = (int)blockIdx.x; t < Tt; t += (int)gridDim.x)
            gemm_tail_tile<EPI, 4, 2, 4>(A, lda, W, bias, M, K, out, ldo, tail_m0 + (t / ttn) * 128, (t % ttn) * 128, lds, tid, wave, lane);
    }
#if MDR_GEMM_ABL == 5
    stamp(6);
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&g_gemm_stamp[i], stamp_sum[i]);
        atomicAdd(&g_gemm_stamp[7], (unsigned long long)total);
    }
#endif
}

using GemmBig = GemmCfg<256, 256, 4, 2, 2>;
using GemmMid = GemmCfg<128, 128, 2, 2, 2>;
using GemmSmall = GemmCfg<64, 64, 2, 2, 3>;   // 3 stages (48 KiB, 3 blocks per CU): 12 % faster than 2 at 2.4 k rows, 4 stages no better

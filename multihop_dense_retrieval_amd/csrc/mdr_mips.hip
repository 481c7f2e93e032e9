// csrc/mdr_mips.hip -- brute-force maximum-inner-product search for gfx950 (MI355X).
//
// Replaces faiss.IndexFlatIP.{add,search} as the reference uses them
// (/root/reference/scripts/eval/eval_mhop_retrieval.py:121-122,155,179). See DESIGN.md §3.
//
// Storage (MDR_STORE_F32X2H). Every fp32 element x is kept as an fp16 pair
//     hi = fp16(x)            lo = fp16((x - hi) * 2^11)          x ~= hi + lo * 2^-11   (22-bit mantissa)
// = 4 bytes per element, the same HBM bytes as fp32, kept as TWO planes (all hi, all lo), each laid out
// in MFMA-operand order so that the HBM->LDS DMA and the LDS->register reads are perfectly linear:
//     plane -> row-block rb (16 rows) -> k-block kb (32 columns) -> 1 KiB fragment block
//     fragment block: lane l = (row & 15) + 16 * ((col & 31) >> 3) holds 8 consecutive columns (16 B)
// which is exactly the A/B operand layout of v_mfma_f32_16x16x32_f16.
//
// Search, k == 1, nq <= 128 per pass ("screen" kernel, the headline path): only the HI plane is
// streamed (2 bytes/element) and scored with ONE fp16 MFMA per k-block,  s_hi = qh.xh.  For every
// (q, x):  |q.x - qh.xh| <= B_q = 1.2e-3 * |q| * max_row|x|   (fp16 rounding of both operands, 2^-10,
// plus fp32 accumulation slack), so a row can only be the winner if  s_hi >= (largest s_hi seen) - 2 B_q.
// Those few rows (a handful per query: the running maxima and near-ties) are appended to a candidate
// list and re-scored exactly from both planes by mips_refine_kernel; if the list overflows (pathological
// all-ties corpora) a flag makes the exact 3-MFMA stream kernel below run instead. Results are identical
// to scoring every row exactly; HBM traffic is half the fp32 matrix.
//
// Search, nq <= 128 per pass ("stream" kernel): one 512-thread workgroup per CU, persistent over
// row-blocks. Wave w keeps the (hi, lo) fragments of queries 16w..16w+15 for ALL of K in registers
// (2 * NKB * 4 VGPRs), the corpus streams HBM -> LDS (global_load_lds, 3-deep ring of 1 row-block
// each) -> every wave's MFMA A operand. Three fp16 MFMAs per (row-block, k-block) give an
// fp32-accurate score:   q.x ~= qh.xh + (qh.xl + ql.xh) * 2^-11     (ql.xl ~ 2^-22 dropped)
// top-1 lives in two registers per lane; top-k (k <= 128) in per-wave candidate lists with a
// running threshold. The [nq, N] score matrix is never materialised.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cfloat>
#include <cstring>
#include <new>

#include "mdr_common.h"

namespace mdr {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short ushort8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

constexpr int kRowBlock = 16;          // corpus rows per MFMA tile
constexpr int kFragBytes = 1024;       // one 16x32 fp16 fragment block
constexpr float kLoScale = 2048.0f;    // 2^11
constexpr float kLoInv = 1.0f / 2048.0f;
constexpr int kStreamQ = 128;          // queries per pass of the stream kernel (8 waves x 16)
constexpr int kStreamCap = 256;        // candidate slots per (workgroup, query) in the stream kernel
constexpr int kGenericQ = 64;          // queries per pass of the generic kernel (4 waves x 16)
constexpr int kGenericCap = 2048;      // >= 2 * MDR_KMAX'
constexpr int kKMax = 1024;            // effective k limit (<= MDR_KMAX)
constexpr int kMergeLds = 6144;        // keys the merge kernel can hold in LDS (48 KiB)

// ---- order-preserving packing: (score desc, row asc)  <=>  key desc -------------------------------
__host__ __device__ inline unsigned ord32(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unord32(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline u64 make_key(float s, unsigned row) { return ((u64)ord32(s) << 32) | (u64)(0xFFFFFFFFu - row); }
__host__ __device__ inline float key_score(u64 k) { return unord32((unsigned)(k >> 32)); }
__host__ __device__ inline unsigned key_row(u64 k) { return 0xFFFFFFFFu - (unsigned)k; }

// ---- fragment-tiled addressing ---------------------------------------------------------------------
__host__ __device__ inline size_t frag_offset(long long row, int col, int nkb) {
    long long rb = row >> 4;
    int rr = (int)(row & 15), kb = col >> 5, g = (col & 31) >> 3, j = col & 7;
    return (size_t)rb * ((size_t)nkb * kFragBytes) + (size_t)kb * kFragBytes + (size_t)(rr + 16 * g) * 16 + (size_t)j * 2;  // within one plane
}

// ---- conversion: row-major {f32,bf16,f16} -> fragment-tiled (hi, lo) fp16 -------------------------
template <typename T>
__device__ inline float load_as_f32(const T* p);
template <>
__device__ inline float load_as_f32<float>(const float* p) { return *p; }
template <>
__device__ inline float load_as_f32<unsigned short>(const unsigned short* p) {  // bf16 bits
    unsigned u = ((unsigned)*p) << 16;
    return __uint_as_float(u);
}
template <>
__device__ inline float load_as_f32<_Float16>(const _Float16* p) { return (float)*p; }

// one thread per (row, 8-column group); rows [n_valid, n_total) are written as zeros (padding)
__device__ inline unsigned short f32_to_bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// BF = false: F32X2H planes (fp16 hi + fp16 lo);  BF = true: one plane of bf16 values (dst_lo unused)
// F32X2H rows are stored as x * 2^-E (xinv = 2^-E, exact) for ONE exponent E per index, chosen by add() so that the largest magnitude it
// has seen sits near 2^9..2^10: fp16's 11 significant bits (22 with the lo plane) then cover the data whatever its absolute scale is --
// rows of magnitude 1e-6 are not lost in fp16 subnormals, rows of magnitude 1e5 do not overflow -- and every score is multiplied
// back by 2^E (exact) where it leaves the library. FAISS IndexFlatIP.add takes any finite fp32 (eval_mhop_retrieval.py:94,122).
template <typename T, bool BF>
__global__ void __launch_bounds__(256) convert_to_frag_kernel(const T* __restrict__ src, long long n_valid, long long n_total,
                                                              int d, long long row0, char* __restrict__ dst_hi, char* __restrict__ dst_lo,
                                                              int* __restrict__ flags, float xinv) {
    const int gpr = d >> 3;
    const int nkb = d >> 5;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long r = idx / gpr;
    int gi = (int)(idx - r * gpr);
    if (r >= n_total) return;
    half8 h, l;
    ushort8 hb;  // bf16 bit patterns (BF): kept in an integer vector, element-wise bit_cast of half8 lanes is avoided
    bool bad = false;
    if (r < n_valid) {
        const T* p = src + r * (long long)d + gi * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = load_as_f32<T>(p + j);
            if (BF) {
                if (!(fabsf(x) <= 3.0e38f)) bad = true;
                hb[j] = f32_to_bf16_rne(x);
            } else {
                if (!(fabsf(x) <= 3.0e38f)) bad = true;
                x *= xinv;
                if (!(fabsf(x) <= 32768.0f)) bad = true;  // (cannot happen for finite x: add() fits E to the data first)
                _Float16 hh = (_Float16)x;
                float res = x - (float)hh;
                h[j] = hh;
                l[j] = (_Float16)(res * kLoScale);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[j] = (_Float16)0.f; l[j] = (_Float16)0.f; hb[j] = 0; }
    }
    if (bad) atomicOr(flags, 1);
    size_t off = frag_offset(row0 + r, gi * 8, nkb);
    if (BF) {
        *(ushort8*)(dst_hi + off) = hb;
    } else {
        *(half8*)(dst_hi + off) = h;
        *(half8*)(dst_lo + off) = l;
    }
}

// one wave per row: flags[2] (as float bits) = max over rows of sum(x^2)   (non-negative floats order like ints)
template <typename T>
__global__ void __launch_bounds__(256) row_norm2_max_kernel(const T* __restrict__ src, long long n, int d, int* __restrict__ flags, float xinv) {
    const int lane = threadIdx.x & 63;
    long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) { float x = load_as_f32<T>(src + r * (long long)d + c) * xinv; s += x * x; }  // in STORED units (x 2^-E)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0 && s == s && __float_as_int(s) > flags[2]) atomicMax(flags + 2, __float_as_int(s));  // pre-check: one hot word
}

// flags[4] (as float bits) = max |x| over the rows an add() is about to take; a NaN / inf leaves a non-finite pattern there
template <typename T>
__global__ void __launch_bounds__(256) absmax_kernel(const T* __restrict__ src, long long count, int* __restrict__ flags) {
    float m = 0.f;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const float x = fabsf(load_as_f32<T>(src + i));
        if (!(x <= 3.0e38f)) bad = true;
        m = fmaxf(m, x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicMax(flags + 4, 0x7F800000);  // +inf: "non-finite seen"
    if ((threadIdx.x & 63) == 0 && __float_as_int(m) > flags[4]) atomicMax(flags + 4, __float_as_int(m));
}
// the stored planes times a power of two (the index exponent E grew): exact unless a value falls below fp16's range
__global__ void __launch_bounds__(256) rescale_planes_kernel(char* __restrict__ hi, char* __restrict__ lo, long long n_vec8, float f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vec8) return;
    half8 h = *(const half8*)(hi + i * 16), l = *(const half8*)(lo + i * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (_Float16)((float)h[j] * f); l[j] = (_Float16)((float)l[j] * f); }
    *(half8*)(hi + i * 16) = h;
    *(half8*)(lo + i * 16) = l;
}

// Query preparation, one wave per query row (rows >= nq are zero padding). Every query is PRE-SCALED by a power of two
//     s = 2^e,  max_i |q_i| / s in [0.5, 1)            (s = 1 for an all-zero row)
// before it is rounded to fp16 / bf16: ranking is invariant to a positive query scale, the division is exact, and the
// MFMA operands are then always in fp16's well-conditioned range whatever the caller's magnitudes are (|q_i| > 65504 would
// otherwise round to inf, tiny queries into subnormals where the relative bound below does not hold). Writes
//   qhi / qlo   fragment-tiled fp16 (hi, lo) pair of q / s (BF: one plane of bf16 bit patterns)
//   qscale[q]   s (the exact stream kernel multiplies its scores back; the screen kernels' scores stay internal, their
//               survivors are re-scored from the caller's fp32 query)
//   bound[q]    B = c * |q / s| * max_row|x| * 1.0001 + 1e-4  >=  |(q/s).x - fp16(q/s).fp16(x)|  for every stored row x:
//               c covers the two operand roundings (2^-10; bf16 rows: the query's 2^-9 only) plus fp32 accumulation; since
//               |q/s| >= 0.5 the relative term also dominates the absolute error of elements that fall into fp16's subnormal
//               range (2^-25 each), for which the 1e-4 is a second belt.
// flags[1] is raised for a non-finite query element (results for that query are unspecified, as with FAISS).
template <bool BF>
__global__ void __launch_bounds__(256) prep_queries_kernel(const float* __restrict__ q, int nq, int nq_pad, int d, int* __restrict__ flags, float c,
                                                           char* __restrict__ qhi, char* __restrict__ qlo, float* __restrict__ bound,
                                                           float* __restrict__ qscale, float xs /* 2^E of the stored rows: folded into qscale */) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq_pad) return;
    const int nkb = d >> 5, ngrp = d >> 3;
    float mx = 0.f, ss = 0.f;
    bool bad = false;
    if (i < nq)
        for (int col = lane; col < d; col += 64) {
            const float x = q[(size_t)i * d + col];
            if (!(fabsf(x) <= 3.0e38f)) bad = true;
            mx = fmaxf(mx, fabsf(x));
            ss = fmaf(x, x, ss);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); ss += __shfl_xor(ss, o); }
    if (__ballot(bad) && lane == 0) atomicOr(flags + 1, 1);
    int e = 0;
    float sc = 1.f;
    if (mx > 0.f && mx <= 3.0e38f) { (void)frexpf(mx, &e); sc = ldexpf(1.f, e); }
    const float inv = 1.f / sc;  // exact: a power of two
    for (int g = lane; g < ngrp; g += 64) {
        half8 h, l;
        ushort8 hb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = i < nq ? q[(size_t)i * d + g * 8 + j] * inv : 0.f;
            if (BF) {
                hb[j] = f32_to_bf16_rne(x);
            } else {
                const _Float16 hh = (_Float16)x;
                h[j] = hh;
                l[j] = (_Float16)((x - (float)hh) * kLoScale);
            }
        }
        const size_t off = frag_offset(i, g * 8, nkb);
        if (BF) {
            *(ushort8*)(qhi + off) = hb;
        } else {
            *(half8*)(qhi + off) = h;
            *(half8*)(qlo + off) = l;
        }
    }
    if (lane == 0) {
        qscale[i] = sc * xs;
        bound[i] = i < nq ? c * (sqrtf(ss) * inv) * sqrtf(__int_as_float(flags[2])) * 1.0001f + 1e-4f : 0.f;
    }
}

__device__ inline int block_sum_256(int v, int* red) {
    // red: LDS int[4]
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

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
// (interleaved A/B of two builds of these sources, gpurun_out r02a; scripts/gpu_ab.sh rebuilds the comparison).
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
                   int* __restrict__ cand_cnt /* [waves] */, int* __restrict__ overflow, const int* __restrict__ run_if = nullptr) {
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
    if (MODE == 2 && n_it > kSampleStagesK) { step_ = n_it / kSampleStagesK; n_it = kSampleStagesK; }
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
    }
    if (MODE == 0) {
        float hm = fmaxf(hmax, __shfl_xor(hmax, 16));
        hm = fmaxf(hm, __shfl_xor(hm, 32));
        if (lane < 16 && q_valid && hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
    } else if (MODE == 2) {  // per-workgroup maxima (no atomics): gmax is [G][kStreamQ] here
        float hm = fmaxf(hmax, __shfl_xor(hmax, 16));
        hm = fmaxf(hm, __shfl_xor(hm, 32));
        if (lane < 16 && q_valid) gmax[(size_t)b * kStreamQ + qlocal] = hm > -FLT_MAX ? ord32(hm) : 0u;
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
                     const int* __restrict__ run_if = nullptr) {
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
    if (MODE == 2 && n_it > kSampleStagesK) { step_ = n_it / kSampleStagesK; n_it = kSampleStagesK; }
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
    }
    if (MODE == 0) {
        const float hm = fmaxf(hmax, __shfl_xor(hmax, 32));
        if (lane < 32 && q_valid && hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
    } else if (MODE == 2) {
        const float hm = fmaxf(hmax, __shfl_xor(hmax, 32));
        if (lane < 32 && q_valid) gmax[(size_t)b * kWideQ + qlocal] = hm > -FLT_MAX ? ord32(hm) : 0u;
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

// ---- the screen kernel for 2 <= k <= 128: hi plane only, per-(workgroup, query) lists keyed by s_hi ---------
// A row can be among the k best exact scores only if s_hi >= h_k - 2B, h_k = k-th largest s_hi over ALL rows
// (k rows have exact >= h_k - B, and s_hi < h_k - 2B means exact < h_k - B). Any lower bound on h_k will do:
//   1. sample pass (mips_screen_kernel MODE 2): every workgroup scores kSampleStagesK super-blocks and publishes its
//      largest s_hi per query; the k-th largest of those G maxima (k distinct rows!) is the first bound (tau0).
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

// k-th largest of the G per-workgroup sample maxima of each query -> tau0 (-inf when fewer than k workgroups saw a row)
__global__ void __launch_bounds__(256) kth_of_maxima_kernel(const unsigned* __restrict__ wgmax /* [G][qcap] ordered, 0 = none */, int G, int k,
                                                            float* __restrict__ tau0 /* [qcap] */, int qcap) {
    const int ql = blockIdx.x;
    __shared__ unsigned v[1024];
    __shared__ int found;
    if (threadIdx.x == 0) found = 0;
    for (int i = threadIdx.x; i < G; i += 256) v[i] = wgmax[(size_t)i * qcap + ql];
    __syncthreads();
    for (int i = threadIdx.x; i < G; i += 256) {
        const unsigned me = v[i];
        if (me == 0u) continue;
        int rank = 0;
        for (int j = 0; j < G; ++j) rank += (v[j] > me) || (v[j] == me && j < i);
        if (rank == k - 1) { tau0[ql] = unord32(me); found = 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && !found) tau0[ql] = -INFINITY;
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
        if (it + 2 < n_it) issue_super_block<NKB>(Xhi, b + (it + 2) * G, lds + ((it + 2) % 3) * SB_BYTES, wave, lane);
        if (!wave_active) continue;

        const char* p = lds + (it % 3) * SB_BYTES + rd_off;
        const f32x16 acc = mfma_chain32<NKB, BF>(p, qf);
        float m16 = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m16 = fmaxf(m16, acc[r]);
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

// ---- generic kernel: any d (multiple of 32), fp32 FMA on the reconstructed values ------------------
// Correctness reference on the device and fallback for shapes the stream kernel does not cover.
template <bool BF>
__global__ void __launch_bounds__(256)
mips_generic_kernel(const char* __restrict__ Xhi, const char* __restrict__ Xlo, long long n_rows, int n_rb, int nkb, const float* __restrict__ q, int nq,
                    u64* __restrict__ cand, int* __restrict__ cand_cnt, u64* __restrict__ cand_kth, int k, const int* __restrict__ run_if, float xs) {
    __shared__ int lds_cnt[kGenericQ];
    if (run_if && *run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = nkb * 32;
    if (threadIdx.x < kGenericQ) lds_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int qlocal = wave * 16 + (lane & 15);
    const bool q_valid = qlocal < nq;
    const bool wave_active = wave * 16 < nq;
    const float* qp = q + (size_t)(q_valid ? qlocal : 0) * d;
    const int g4 = lane >> 4;
    float tau = -INFINITY;
    u64* wave_lists = cand + ((size_t)blockIdx.x * kGenericQ + (size_t)wave * 16) * kGenericCap;
    u64* my_list = wave_lists + (size_t)(lane & 15) * kGenericCap;
    const size_t rb_bytes = (size_t)nkb * kFragBytes;
    if (wave_active) {
        for (int rb = blockIdx.x; rb < n_rb; rb += gridDim.x) {
            const size_t blk = (size_t)rb * rb_bytes;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb < nkb; ++kb) {
                for (int gp = 0; gp < 4; ++gp) {
                    float qv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) qv[j] = qp[kb * 32 + gp * 8 + j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t e = blk + (size_t)kb * kFragBytes + (size_t)((4 * g4 + r) + 16 * gp) * 16;
                        if (BF) {
                            const ushort8 hb = *(const ushort8*)(Xhi + e);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[r] = fmaf(bf16_bits_to_f32(hb[j]), qv[j], acc[r]);
                        } else {
                            half8 h = *(const half8*)(Xhi + e);
                            half8 l = *(const half8*)(Xlo + e);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[r] = fmaf((float)h[j] + (float)l[j] * kLoInv, qv[j], acc[r]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned row = (unsigned)rb * 16u + 4u * g4 + r;
                consider(acc[r] * xs, row, ((long long)row < n_rows) && q_valid, tau, my_list, lds_cnt + qlocal);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wave_prune_if_needed<kGenericCap / 64, kGenericCap>(wave_lists, lds_cnt + wave * 16, k, lane, tau, false, nullptr);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_prune_if_needed<kGenericCap / 64, kGenericCap>(wave_lists, lds_cnt + wave * 16, k, lane, tau, true,
                                                            cand_kth + (size_t)blockIdx.x * kGenericQ + wave * 16);
        if (lane < 16) cand_cnt[(size_t)blockIdx.x * kGenericQ + qlocal] = lds_cnt[qlocal];
    }
}

// ---- result kernels ----------------------------------------------------------------------------------
// =====================================================================================================================
// int8 screening tier (k = 1, F32X2H storage): HALF the bytes of the fp16 hi plane per corpus pass.
// Every row is stored a third time as int8 with its own scale, x_i = s_r (x8_i + e_i), |e_i| <= 1/2 (s_r = max_i|x_i| / 127),
// every query is quantised the same way, q_i = t (q8_i + f_i), |f_i| <= 1/2, and v_mfma_i32_16x16x64_i8 accumulates
// A = sum q8_i x8_i exactly. Then
//     q.x = t s_r (A + sum q8_i e_i + sum f_i x8_i + sum f_i e_i),   |q.x - t s_r A| <= t s_r (L1(q8)/2 + L1(x8_r)/2 + d/4)
// so with  alpha_q = t L1(q8) / 2  and  beta_q = t max_r [ s_r (L1(x8_r)/2 + d/4) ]  (the max is kept by add(), i8stats[1])
//     L_r = s_r (t A - alpha_q) - beta_q  <=  q.x_r  <=  s_r (t A + alpha_q) + beta_q = U_r
// (both inflated by 1e-3 for the fp32 roundings of the scales and of these two FMAs). A row can be the best row only if
// U_r >= max_r' L_r'; the kernel keeps the running maximum of the lower bounds (`known`) exactly the way mips_screen_kernel keeps its
// running s_hi, appends rows with U_r >= known to the same per-wave candidate lists, and mips_refine_kernel re-scores them from
// the fp16 (hi, lo) planes: ids and scores are those of the exact path. If a list overflows (data for which the int8 bound is
// loose: a large common mean, very heavy tails) the fp16 screen runs behind it, and the exact pass behind that -- each
// skipped on the device when the tier before it did not overflow.
// Layout: a super-block (32 rows) = 2 x NKB8 fragment blocks of 1 KiB (16 rows x 64 int8; lane (lr, g) of the MFMA owns the
// 16 bytes k = 64 kb + 16 g .. of row lr at (16 g + lr) * 16) followed by 256 bytes holding the 32 row scales: 24.25 KiB at
// d = 768 against the 48 KiB of the fp16 hi plane.
#ifndef MDR_I8_ABL
#define MDR_I8_ABL 0  // measurement builds (wrong results): 1 no scale-tail DMA, 2 no epilogue, 3 no MFMAs, 4 no fragment reads
#endif
constexpr int kI8RefinePerQuery = 8192;  // emitted candidates per query of a pass beyond which the int8 tier hands over to the fp16 screen (see mips_refine8_kernel)
constexpr int kI8Tail = 256;  // bytes behind a super-block's fragments: 32 fp32 row scales (+ padding to one 4-byte-per-lane DMA piece)
#ifndef MDR_I8_ALIGN
#define MDR_I8_ALIGN 256  // variant-build knob: alignment of a super-block's start in the int8 plane
#endif
__host__ __device__ inline size_t i8_sb_bytes(int nkb8) { return ((size_t)2 * nkb8 * kFragBytes + kI8Tail + MDR_I8_ALIGN - 1) / MDR_I8_ALIGN * MDR_I8_ALIGN; }
typedef int i32x4 __attribute__((ext_vector_type(4)));
// candidate of the int8 tier: query (16 bits) | upper bound U as the top 16 bits of its ordered representation, rounded UP | row
__device__ inline u64 pack_cand8(int qi, float u, unsigned row) {
    unsigned o = ord32(u);
    o = o > 0xFFFF0000u ? 0xFFFFu : (o + 0xFFFFu) >> 16;
    return ((u64)(unsigned)qi << 48) | ((u64)o << 32) | row;
}

__device__ inline float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Centre of the int8 plane. Real embedding matrices are anisotropic (LayerNorm outputs share a bias vector and a few large, row-independent
// coordinates): quantised as they are, one outlier coordinate sets every row scale s_r and the bounds widen by its size. Since
//     q.x = q.(x - c) + q.c          and q.c is the same for every row of a query,
// the plane stores x - c for a fixed vector c and the screen ranks rows by bounds on q.(x - c): identical ranking, bounds as tight as for
// centred data. c = the column means of (at most the first 65536 rows of) the FIRST add(), frozen afterwards -- ANY fixed c is correct,
// a good one is only faster. Where a centred bound meets an exact (uncentred) score -- the `known` seeds and the thresholds of
// mips_refine8_kernel -- the per-query offset q.c (+ its fp32 rounding slack: qab[q][3]) is subtracted from the exact score first.
// The same identity holds coordinate by coordinate for any positive weights w:  q.(x - c) = sum_i (q_i w_i) ((x_i - c_i) / w_i).  Outlier
// coordinates of real embeddings are large but nearly CONSTANT across rows; with w_i = the column's standard deviation the plane stores
// (x_i - c_i) / w_i ~ unit variance in every coordinate and the query enters as q_i w_i, so a query's own outlier coordinate (which would
// otherwise set its quantisation step t for all 768 coordinates) shrinks to the size of the others. w is taken with c and frozen with it;
// it is a power of two (exact scaling) clamped to [2^-12, 2^12] times the median-free reference 1 (a constant column gets w = 1).
// One block per 64 columns; block (x, y): rows y, y + gridDim.y, ...; partial sums / sums of squares are combined with atomicAdd into
// a zeroed buffer (sums[0..d) and sums[d..2d)).
template <typename T>
__global__ void __launch_bounds__(256) col_sum_kernel(const T* __restrict__ src, long long n, int d, float* __restrict__ sums) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rsub = threadIdx.x >> 6;
    float a1 = 0.f, a2 = 0.f;
    if (c < d)
        for (long long r = (long long)blockIdx.y * 4 + rsub; r < n; r += (long long)gridDim.y * 4) {
            const float x = load_as_f32<T>(src + r * (long long)d + c);
            a1 += x;
            a2 = fmaf(x, x, a2);
        }
    red[0][rsub][threadIdx.x & 63] = a1;
    red[1][rsub][threadIdx.x & 63] = a2;
    __syncthreads();
    if (rsub == 0 && c < d) {
        atomicAdd(sums + c, red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
        atomicAdd(sums + d + c, red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
    }
}
// sums -> cw[0..d) = centre (column means), cw[d..2d) = 1 / w, cw[2d..3d) = w.
// The quantisation steps are set by the LARGEST scaled coordinate on either side: s_r ~ max_i |x_i - c_i| / w_i for the rows and
// t ~ max_i |q_i| w_i for a query, and the bound is about s_r |q w|_1 / 2 + t |(x - c) / w|_1 / 2. With A_i = std_i / ref (a column's spread
// relative to the typical spread ref = RMS of the column stds) and B_i = (|c_i| + 3.5 std_i) / (3.5 ref) (how large a QUERY's coordinate is
// expected to be there: queries are embeddings of the same kind as the rows), the weights that minimise X + Y = max_i A_i / w_i + max_i B_i w_i
// are any w_i in [A_i / X, X / B_i] with X = Y = sqrt(max(1, max_i A_i B_i)); w_i = 1 wherever that interval contains 1 (isotropic
// data: everywhere), the nearer end otherwise, rounded to a power of two (exact scaling) in 2^+-12. A large, nearly constant outlier
// coordinate (A small, B large) is scaled DOWN so that the query's outlier shrinks while the rows' small spread there still resolves; a
// dense common mean needs nothing on the row side (the centre removes it) and a little on the query side.
// One block of 1024 threads (d <= 1024).
__global__ void __launch_bounds__(1024) centre_finish_kernel(const float* __restrict__ sums, int d, float inv_n, float* __restrict__ cw) {
    __shared__ float red[16];
    __shared__ float bc;
    const int i = threadIdx.x;
    auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float u = __shfl_xor(v, o); v = is_max ? fmaxf(v, u) : v + u; }
        __syncthreads();
        if ((i & 63) == 0) red[i >> 6] = v;
        __syncthreads();
        if (i == 0) {
            float t = red[0];
            for (int k = 1; k < 16; ++k) t = is_max ? fmaxf(t, red[k]) : t + red[k];
            bc = t;
        }
        __syncthreads();
        return bc;
    };
    float mu = 0.f, var = 0.f;
    if (i < d) {
        mu = sums[i] * inv_n;
        var = fmaxf(sums[d + i] * inv_n - mu * mu, 0.f);
        if (!(fabsf(mu) <= 3.0e38f) || !(var <= 3.0e38f)) { mu = 0.f; var = 0.f; }  // (non-finite rows: the add is rejected anyway; keep c and w finite)
    }
    const float ref = sqrtf(block_reduce(var, false) / (float)d);
    float A = 0.f, B = 0.f;
    if (i < d && ref > 0.f) {
        const float sd = sqrtf(var);
        A = sd / ref;
        B = (fabsf(mu) + 3.5f * sd) / (3.5f * ref);
    }
    const float X = sqrtf(fmaxf(1.f, block_reduce(A * B, true)));
    if (i >= d) return;
    float w = 1.f;
    if (ref > 0.f) {
        const float lo = A / X, hi = B > 0.f ? X / B : 3.0e38f;  // lo <= hi because A B <= X^2
        const float wr = lo > 1.f ? lo : (hi < 1.f ? hi : 1.f);
        int e = (int)rintf(log2f(fmaxf(wr, 1e-30f)));
        e = e < -12 ? -12 : (e > 12 ? 12 : e);
        w = ldexpf(1.f, e);
    }
    cw[i] = mu;
    cw[d + i] = 1.f / w;
    cw[2 * d + i] = w;
}

// one wave per row: lanes 0 .. d/16-1 quantise 16 consecutive columns each of x - centre. stats[0] = max s_r, stats[1] = max s_r (L1(x8_r)/2 + d/4)
// (non-negative floats, kept as their bit patterns: they order like ints)
template <typename T>
__global__ void __launch_bounds__(256) convert_to_i8_kernel(const T* __restrict__ src, long long n, int d, long long row0, char* __restrict__ dst,
                                                            int* __restrict__ stats, const float* __restrict__ centre) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int nkb8 = d >> 6;
    const bool on = lane < (d >> 4);
    float x[16];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x[j] = on ? (load_as_f32<T>(src + r * (long long)d + lane * 16 + j) - centre[lane * 16 + j]) * centre[d + lane * 16 + j] : 0.f;  // (x - c) / w
        mx = fmaxf(mx, fabsf(x[j]));
    }
    mx = wave_max_f(mx);
    const float sc = mx > 0.f ? mx / 127.f : 0.f;
    const float inv = mx > 0.f ? 127.f / mx : 0.f;
    int l1 = 0;
    i32x4 packed;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned u = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int v = (int)rintf(x[4 * w + b] * inv);
            v = v > 127 ? 127 : (v < -127 ? -127 : v);
            l1 += v < 0 ? -v : v;
            u |= ((unsigned)v & 0xFFu) << (8 * b);
        }
        packed[w] = (int)u;
    }
    l1 = wave_sum_i(l1);
    const long long row = row0 + r;
    char* sb = dst + (size_t)(row >> 5) * i8_sb_bytes(nkb8);
    if (on) {
        const int kb = lane >> 2, g = lane & 3;
        *(i32x4*)(sb + ((size_t)((row >> 4) & 1) * nkb8 + kb) * kFragBytes + (g * 16 + (int)(row & 15)) * 16) = packed;
    }
    if (lane == 0) {
        *(float*)(sb + (size_t)2 * nkb8 * kFragBytes + (row & 31) * 4) = sc;
        const float c = sc * (0.5f * (float)l1 + 0.25f * (float)d);
        if (__float_as_int(sc) > stats[0]) atomicMax(stats + 0, __float_as_int(sc));
        if (__float_as_int(c) > stats[1]) atomicMax(stats + 1, __float_as_int(c));
    }
}

// one wave per query row (rows >= nq: zero padding). q8: fragment-tiled like the corpus blocks (16 queries per block, NKB8 KiB
// each); qab[i] = (t, alpha, beta, 0) with the 1e-3 inflation described above.
// qab[i][3] = q.c + slack (c = the plane's centre): what is subtracted from an EXACT score of a row to get a valid lower bound of its
// centred score q.(x - c). slack = 1e-4 sum|q_i c_i| + 2e-6 |q.c| covers the fp32 summation of q.c (768 terms) and the fp32 rounding of
// x - c in convert_to_i8_kernel; the 1e-3 inflation of alpha / beta covers the rest as before.
__global__ void __launch_bounds__(256) prep_queries_i8_kernel(const float* __restrict__ q, int nq, int nq_pad, int d, const int* __restrict__ stats,
                                                              char* __restrict__ q8, f32x4* __restrict__ qab, const float* __restrict__ centre) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq_pad) return;
    const int nkb8 = d >> 6;
    const bool on = lane < (d >> 4) && i < nq;
    float x[16];
    float mx = 0.f, qc = 0.f, qca = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x[j] = on ? q[(size_t)i * d + lane * 16 + j] : 0.f;
        const float cj = on ? centre[lane * 16 + j] : 0.f;
        qc = fmaf(x[j], cj, qc);
        qca = fmaf(fabsf(x[j]), fabsf(cj), qca);
        x[j] *= on ? centre[2 * d + lane * 16 + j] : 0.f;  // q_i w_i (w a power of two: exact)
        mx = fmaxf(mx, fabsf(x[j]));
    }
    mx = wave_max_f(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qc += __shfl_xor(qc, o); qca += __shfl_xor(qca, o); }
    const bool fin = mx <= 3.0e38f;  // a non-finite query gets an infinite bound below: every row becomes a candidate, the lists overflow, the tiers behind decide
    const float t = mx > 0.f && fin ? mx / 127.f : 0.f;
    const float inv = mx > 0.f && fin ? 127.f / mx : 0.f;
    int l1 = 0;
    i32x4 packed;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned u = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int v = (int)rintf(x[4 * w + b] * inv);
            v = v > 127 ? 127 : (v < -127 ? -127 : v);
            l1 += v < 0 ? -v : v;
            u |= ((unsigned)v & 0xFFu) << (8 * b);
        }
        packed[w] = (int)u;
    }
    l1 = wave_sum_i(l1);
    if (lane < (d >> 4)) {
        const int kb = lane >> 2, g = lane & 3;
        *(i32x4*)(q8 + ((size_t)(i >> 4) * nkb8 + kb) * kFragBytes + (g * 16 + (i & 15)) * 16) = packed;
    }
    if (lane == 0) {
        const float s2 = __int_as_float(stats[1]);
        f32x4 o = {t, 0.5f * t * (float)l1 * 1.001f, t * s2 * 1.001f, qc + (1e-4f * qca + 2e-6f * fabsf(qc))};
        if (!fin) o = (f32x4){0.f, INFINITY, INFINITY, 0.f};
        qab[i] = o;
    }
}

template <int NKB8>
__device__ __forceinline__ void issue_super_block8(const char* __restrict__ X8, int sb, char* slot, int wave, int lane) {
    constexpr int CPW = NKB8 / 4;  // 2 * NKB8 fragment pieces over 8 waves
    const char* g = X8 + (size_t)sb * i8_sb_bytes(NKB8) + (size_t)wave * CPW * kFragBytes + lane * 16;
    char* l = slot + wave * CPW * kFragBytes;
#pragma unroll
    for (int c = 0; c < CPW; ++c) __builtin_amdgcn_global_load_lds(MDR_GPTR(g + c * kFragBytes), MDR_LPTR(l + c * kFragBytes), 16, 0, MDR_MIPS_DMA_AUX);
    if (wave == 0 && MDR_I8_ABL != 1)  // the scale tail: one 4-byte-per-lane piece
        __builtin_amdgcn_global_load_lds(MDR_GPTR(X8 + (size_t)sb * i8_sb_bytes(NKB8) + 2 * NKB8 * kFragBytes + lane * 4),
                                         MDR_LPTR(slot + 2 * NKB8 * kFragBytes), 4, 0, MDR_MIPS_DMA_AUX);
}

// MODE 0: sample pass (publish the largest lower bound per query to gmax); MODE 1: main pass (candidates). See mips_screen_kernel.
template <int NKB8, int MODE, int NS>  // NS: LDS slots of one super-block (NS - 1 stages in flight)
__global__ void __launch_bounds__(512, 2)
mips_screen8_kernel(const char* __restrict__ X8, long long n_rows, int n_sb, const char* __restrict__ Q8, const f32x4* __restrict__ qab, int nq, int q_base,
                    unsigned* __restrict__ gmax /* [nq] ordered(max L) */, u64* __restrict__ cand /* [waves][kWaveCandCap] */,
                    int* __restrict__ cand_cnt /* [waves] */, int* __restrict__ overflow, u64* __restrict__ gstar /* [nq] (ordered max L, its row) */,
                    const u64* __restrict__ best /* MODE 1: exact keys of the sample pass's star rows (a tighter first `known`) */) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB_BYTES = 2 * NKB8 * kFragBytes + kI8Tail;
    constexpr int CPW = NKB8 / 4;  // DMA pieces per wave and stage (wave 0: + 1, the scale tail)
    constexpr int HK = NKB8 / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    int n_it = (n_sb - b + G - 1) / G;  // >= 1 (grid <= n_sb)
    if (MODE == 0) {  // the sample pass scores 1/16 of the stages, at most kSampleStages (small shards: fewer)
        const int samp = max(1, min(kSampleStages, n_it >> 4));
        if (n_it > samp) n_it = samp;
    }

#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < n_it) issue_super_block8<NKB8>(X8, b + i * G, lds + i * SB_BYTES, wave, lane);

    const bool wave_active = wave * 16 < nq;
    i32x4 qh[NKB8];
    {
        const size_t qoff = (size_t)wave * NKB8 * kFragBytes + lane * 16;
#pragma unroll
        for (int kb = 0; kb < NKB8; ++kb) qh[kb] = *(const i32x4*)(Q8 + qoff + kb * kFragBytes);
    }
    const int qlocal = wave * 16 + (lane & 15);
    const bool q_valid = qlocal < nq;
    f32x4 ab = {0.f, 0.f, 0.f, 0.f};
    if (q_valid) ab = qab[qlocal];
    float qt = ab[0], qa = ab[1], qb = ab[2];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float known = -FLT_MAX;  // largest lower bound (of the CENTRED score q.(x - c)) known for this lane's query
    if (MODE == 1 && q_valid) {
        unsigned g = gmax[qlocal];
        if (g) known = unord32(g);
        const u64 kb = best[q_base + qlocal];  // the exact score of a real row, minus q.c (+ slack), is a lower bound of the best centred score too
        if (kb) known = fmaxf(known, key_score(kb) - ab[3]);
    }
    // retire every register load before the loop (see mips_screen_kernel)
#pragma unroll
    for (int kb = 0; kb < NKB8; ++kb) asm volatile("" : "+v"(qh[kb]));
    asm volatile("" : "+v"(qt), "+v"(qa), "+v"(qb), "+v"(known));
    const f32x2 qt2 = {qt, qt}, qa2 = {qa, qa}, qb2 = {qb, qb};
    const unsigned sub_row = 4u * (unsigned)(lane >> 4);
    float lmax = -FLT_MAX;  // largest lower bound this lane has seen
    unsigned lrow = 0;      // ... and the row it belongs to: the refinement re-scores that row first (see mips_star8_kernel)
    int my_cnt = 0;
    u64* my_list = cand + ((size_t)b * 8 + wave) * kWaveCandCap;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    for (int it = 0; it < n_it; ++it) {
        if (it + NS - 2 < n_it) {  // NS - 2 younger stages may stay in flight (the last few iterations simply drain)
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((CPW + (MDR_I8_ABL != 1)) * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW * (NS - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 1 && ((it + b) & 31) == 31 && wave_active) {  // exchange lower bounds with the other workgroups (placement: see mips_screen_kernel)
            float hm = fmaxf(lmax, __shfl_xor(lmax, 16));
            hm = fmaxf(hm, __shfl_xor(hm, 32));
            float kn = known;
            if (lane < 16 && q_valid) {
                if (hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
                unsigned g = load_u32_l2(gmax + qlocal);
                if (g) kn = fmaxf(kn, unord32(g));
            }
            known = __shfl(kn, lane & 15);
        }
        if (it + NS - 1 < n_it) issue_super_block8<NKB8>(X8, b + (it + NS - 1) * G, lds + ((it + NS - 1) % NS) * SB_BYTES, wave, lane);
        if (!wave_active || MDR_I8_ABL == 5) continue;

        {
        const int sb_idx = b + it * G;
        const char* slot = lds + (it % NS) * SB_BYTES;
        const char* p = slot + lane * 16;
        i32x4 a00 = {0, 0, 0, 0}, a01 = a00, a10 = a00, a11 = a00;
        constexpr int PF = 2;
        i32x4 x00[PF], x01[PF], x10[PF], x11[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (MDR_I8_ABL == 4) { x00[i] = x01[i] = x10[i] = x11[i] = qh[i]; continue; }
            x00[i] = *(const i32x4*)(p + i * kFragBytes);
            x01[i] = *(const i32x4*)(p + (HK + i) * kFragBytes);
            x10[i] = *(const i32x4*)(p + (NKB8 + i) * kFragBytes);
            x11[i] = *(const i32x4*)(p + (NKB8 + HK + i) * kFragBytes);
        }
        // this lane's 8 row scales: rows 4 g .. 4 g + 3 of both 16-row blocks
        const f32x4 sr0 = *(const f32x4*)(slot + 2 * NKB8 * kFragBytes + sub_row * 4);
        const f32x4 sr1 = *(const f32x4*)(slot + 2 * NKB8 * kFragBytes + (16 + sub_row) * 4);
#pragma unroll
        for (int kb = 0; kb < HK; ++kb) {
            const i32x4 c00 = x00[kb % PF], c01 = x01[kb % PF], c10 = x10[kb % PF], c11 = x11[kb % PF];
            if (kb + PF < HK && MDR_I8_ABL != 4) {
                x00[kb % PF] = *(const i32x4*)(p + (kb + PF) * kFragBytes);
                x01[kb % PF] = *(const i32x4*)(p + (HK + kb + PF) * kFragBytes);
                x10[kb % PF] = *(const i32x4*)(p + (NKB8 + kb + PF) * kFragBytes);
                x11[kb % PF] = *(const i32x4*)(p + (NKB8 + HK + kb + PF) * kFragBytes);
            }
            if (MDR_I8_ABL == 3) { a00 += c00; a01 += c01; a10 += c10; a11 += c11; continue; }
            a00 = __builtin_amdgcn_mfma_i32_16x16x64_i8(c00, qh[kb], a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_i32_16x16x64_i8(c01, qh[HK + kb], a01, 0, 0, 0);
            a10 = __builtin_amdgcn_mfma_i32_16x16x64_i8(c10, qh[kb], a10, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_i32_16x16x64_i8(c11, qh[HK + kb], a11, 0, 0, 0);
        }
        const i32x4 i0 = a00 + a01, i1 = a10 + a11;
        if (MDR_I8_ABL == 2) { if (q_valid) lmax = fmaxf(lmax, (float)(i0[0] + i1[0] + i0[1] + i1[1] + i0[2] + i1[2] + i0[3] + i1[3])); continue; }
        // upper bounds U = s_r (t A + alpha) + beta, two per packed FMA; the lower bound is only needed as a maximum, and
        // max_r L_r >= L_(argmax U) = max U - 2 (alpha s_(argmax U) + beta) >= max U - 2 (alpha max_r s_r + beta)
        f32x2 u2[4];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const f32x2 f0 = {(float)i0[2 * pr], (float)i0[2 * pr + 1]}, f1 = {(float)i1[2 * pr], (float)i1[2 * pr + 1]};
            const f32x2 s0 = {sr0[2 * pr], sr0[2 * pr + 1]}, s1 = {sr1[2 * pr], sr1[2 * pr + 1]};
            u2[pr] = __builtin_elementwise_fma(__builtin_elementwise_fma(f0, qt2, qa2), s0, qb2);
            u2[2 + pr] = __builtin_elementwise_fma(__builtin_elementwise_fma(f1, qt2, qa2), s1, qb2);
        }
        const float up[8] = {u2[0][0], u2[0][1], u2[1][0], u2[1][1], u2[2][0], u2[2][1], u2[3][0], u2[3][1]};
        const unsigned row0 = (unsigned)sb_idx * 32u + sub_row;
        const bool whole = (long long)sb_idx * 32 + 32 <= n_rows;  // wave-uniform
        const float mu = fmaxf(fmaxf(fmaxf(up[0], up[1]), fmaxf(up[2], up[3])), fmaxf(fmaxf(up[4], up[5]), fmaxf(up[6], up[7])));
        const float smax = fmaxf(fmaxf(fmaxf(sr0[0], sr0[1]), fmaxf(sr0[2], sr0[3])), fmaxf(fmaxf(sr1[0], sr1[1]), fmaxf(sr1[2], sr1[3])));
        if (whole && (MODE != 1 || __ballot(q_valid && mu >= known) == 0ull)) {
            const float cl = mu - 2.f * fmaf(qa, smax, qb);
            if (q_valid && cl > lmax) {  // a new record for this lane (O(log rows) times per pass): remember the row
                lmax = cl;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (up[e] == mu) lrow = row0 + 16u * (e >> 2) + (e & 3);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned row = row0 + 16u * h + r;
                    const bool ok = (long long)row < n_rows && q_valid;
                    const float lr_ = up[4 * h + r] - 2.f * fmaf(qa, h ? sr1[r] : sr0[r], qb);
                    if (ok && lr_ > lmax) { lmax = lr_; lrow = row; }
                    if (MODE == 1) {
                        const bool hit = ok && up[4 * h + r] >= known;
                        const u64 m = __ballot(hit);
                        if (m) {  // wave-uniform
                            const int slot_i = my_cnt + __popcll(m & lt);
                            if (hit && slot_i < kWaveCandCap) my_list[slot_i] = pack_cand8(q_base + qlocal, up[4 * h + r], row);
                            my_cnt += __popcll(m);
                        }
                    }
                }
        }
        if (MODE == 1) {  // share the maximum between the 4 lanes of a query
            float hm = fmaxf(lmax, __shfl_xor(lmax, 16));
            hm = fmaxf(hm, __shfl_xor(hm, 32));
            known = fmaxf(known, hm);
        }
        }
    }
    {   // both modes publish: after the main pass gmax holds the largest lower bound over ALL rows, which lets the refinement drop
        // the candidates that were emitted against an early, loose `known`; gstar also names the row that bound belongs to
        float hm = fmaxf(lmax, __shfl_xor(lmax, 16));
        hm = fmaxf(hm, __shfl_xor(hm, 32));
        if (lane < 16 && q_valid && hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
        if (q_valid && lmax == hm && hm > -FLT_MAX) atomicMax(gstar + qlocal, ((u64)ord32(lmax) << 32) | lrow);
    }
    if (MODE == 1 && lane == 0) {
        cand_cnt[b * 8 + wave] = my_cnt < kWaveCandCap ? my_cnt : kWaveCandCap;
        if (my_cnt > kWaveCandCap) *overflow = 1;
        if (my_cnt) atomicAdd(overflow + 3, my_cnt);  // ctl8[3]: candidates emitted by this pass (the refinement's guard)
    }
}

// ---- the int8 tier with 32 queries per wave (256 per pass): v_mfma_i32_32x32x32_i8 ------------------------------------------
// mips_screen32_kernel's tile on the int8 plane: K-slice s (32 columns) of the 32-row super-block, lane (row = l & 31, k = 32 s +
// 16 (l >> 5) ..) sits at ((l >> 4) & 1) * NKB8 KiB + (s >> 1) KiB + (s & 1) * 512 + (l >> 5) * 256 + (l & 15) * 16 of the image
// (the 16x16x64 fragment layout read with the other address pattern, conflict-free); 24 MFMAs per super-block instead of 48,
// 96 registers of resident query slices instead of 192. Reads and their counted waits are hand-placed as in mfma_chain32.
typedef int i32x16 __attribute__((ext_vector_type(16)));

// `fill(sl)` is called behind MFMA sl and pinned there: the caller's VALU work (the epilogue of the PREVIOUS super-block) issues in
// the shadow of the 32-cycle MFMAs instead of after the chain.
#ifndef MDR_I8W_PF
#define MDR_I8W_PF 4  // fragment reads in flight ahead of the MFMA that consumes them (<= 8)
#endif
template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {  // fn(std::integral_constant<int, I>{}) for I = I .. N-1: indices stay compile-time constants
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}

template <int NKB8, typename F>
__device__ __forceinline__ i32x16 mfma_chain8x32(const char* p, const i32x4 (&qf)[2 * NKB8], F&& fill) {
    constexpr int NSL = 2 * NKB8, PF = MDR_I8W_PF;
    const unsigned a = (unsigned)(uintptr_t)p;
    i32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    i32x4 xa[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[i]) : "v"(a), "n"((i >> 1) * kFragBytes + (i & 1) * 512));
    static_for<0, NSL>([&](auto slc) __attribute__((always_inline)) {
        constexpr int sl = decltype(slc)::value;
        constexpr int left = NSL - 1 - sl < PF - 1 ? NSL - 1 - sl : PF - 1;  // reads younger than the one needed now
        if constexpr (left == 7) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(xa[sl % PF]));
        else if constexpr (left == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xa[sl % PF]));
        else if constexpr (left == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(xa[sl % PF]));
        else if constexpr (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xa[sl % PF]));
        else if constexpr (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xa[sl % PF]));
        else if constexpr (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xa[sl % PF]));
        else if constexpr (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xa[sl % PF]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[sl % PF]));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[sl % PF], qf[sl], acc, 0, 0, 0);
        if constexpr (sl + PF < NSL)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[sl % PF]) : "v"(a), "n"(((sl + PF) >> 1) * kFragBytes + ((sl + PF) & 1) * 512));
        fill(slc);
        __builtin_amdgcn_sched_barrier(0);
    });
    return acc;
}

// MDR_I8_ABL=9 builds: s_memtime timeline of wave 0 of every workgroup of the MODE 1 wide kernel, summed:
// [0] wait + barrier, [1] exchange + DMA issue, [2] scale reads + MFMA chain, [3] epilogue (incl. bound sharing), [7] stages
#if MDR_I8_ABL == 9
__device__ unsigned long long g_i8_stamp[8];
#endif

template <int NKB8, int MODE, int NS>
__global__ void __launch_bounds__(512, 2)
mips_screen8w_kernel(const char* __restrict__ X8, long long n_rows, int n_sb, const char* __restrict__ Q8, const f32x4* __restrict__ qab, int nq, int q_base,
                     unsigned* __restrict__ gmax, u64* __restrict__ cand, int* __restrict__ cand_cnt, int* __restrict__ overflow, u64* __restrict__ gstar,
                     const u64* __restrict__ best) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SB_BYTES = 2 * NKB8 * kFragBytes + kI8Tail;
    constexpr int SPS = 2;                   // super-blocks per stage: ONE barrier and one burst of DMA issue per 64 rows
    constexpr int ST_BYTES = SPS * SB_BYTES;
    constexpr int CPW = SPS * (NKB8 / 4);    // DMA pieces per wave and stage (wave 0: + SPS scale tails)
    constexpr int NSL = 2 * NKB8;  // 32-deep K slices
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    const int n_st = (n_sb + SPS - 1) / SPS;  // (the plane is allocated to a whole number of stages)
    int n_it = (n_st - b + G - 1) / G;
    if (MODE == 0) {  // the sample pass scores 1/16 of the stages, at most kSampleStages (small shards: fewer)
        const int samp = max(1, min(kSampleStages, n_it >> 4));
        if (n_it > samp) n_it = samp;
    }
    auto issue_stage = [&](int stg, char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < SPS; ++h) issue_super_block8<NKB8>(X8, SPS * stg + h, dst + h * SB_BYTES, wave, lane);
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < n_it) issue_stage(b + i * G, lds + i * ST_BYTES);

    const bool wave_active = wave * 32 < nq;
    const int l31 = lane & 31, lh = lane >> 5;
    i32x4 qf[NSL];
    {
        const size_t qrow = (size_t)wave * 32 + l31;
        const char* qp = Q8 + (qrow >> 4) * ((size_t)NKB8 * kFragBytes) + (qrow & 15) * 16 + lh * 256;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) qf[sl] = *(const i32x4*)(qp + (sl >> 1) * kFragBytes + (sl & 1) * 512);
    }
    const int qlocal = wave * 32 + l31;
    const bool q_valid = qlocal < nq;
    f32x4 ab = {0.f, 0.f, 0.f, 0.f};
    if (q_valid) ab = qab[qlocal];
    float qt = ab[0], qa = ab[1], qb = ab[2];
    float known = -FLT_MAX;
    if (MODE == 1 && q_valid) {
        unsigned g = gmax[qlocal];
        if (g) known = unord32(g);
        const u64 kb = best[q_base + qlocal];
        if (kb) known = fmaxf(known, key_score(kb) - ab[3]);  // exact score -> centred lower bound (see mips_screen8_kernel)
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) asm volatile("" : "+v"(qf[sl]));
    asm volatile("" : "+v"(qt), "+v"(qa), "+v"(qb), "+v"(known));
    const f32x2 qt2 = {qt, qt}, qa2 = {qa, qa}, qb2 = {qb, qb};
    float lmax = -FLT_MAX;
    unsigned lrow = 0;  // the row lmax belongs to (see mips_screen8_kernel)
    int my_cnt = 0;
    u64* my_list = cand + ((size_t)b * 8 + wave) * kWaveCandCap;
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int rd_off = ((lane >> 4) & 1) * (NKB8 * kFragBytes) + lh * 256 + (lane & 15) * 16;

    // The epilogue of a super-block runs INSIDE the MFMA chain of the next one (software pipelining within the wave): `bounds` turns
    // two accumulators of the other super-block into upper bounds, `decide` tests them; mfma_chain8x32 calls them behind its first
    // nine MFMAs. The two super-blocks of a stage own one accumulator set each (P[0], P[1]), so nothing is copied and the chain
    // of one never waits for the last MFMA of the other to drain.
    struct Pending {
        i32x16 acc;
        f32x4 sr[4];
        int sb;
        bool have;
    };
    Pending P[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        P[h].have = false;
        P[h].sb = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) P[h].acc[e] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) P[h].sr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x2 u2[8];
    float mu = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < 8; ++j) u2[j] = (f32x2){0.f, 0.f};
    auto bounds = [&](const Pending& R, auto jc) __attribute__((always_inline)) {  // accumulators 2 j, 2 j + 1 of super-block R
        constexpr int j = decltype(jc)::value;
        const f32x2 f = {(float)R.acc[2 * j], (float)R.acc[2 * j + 1]};
        const f32x2 sc = {R.sr[j >> 1][2 * (j & 1)], R.sr[j >> 1][2 * (j & 1) + 1]};
        u2[j] = __builtin_elementwise_fma(__builtin_elementwise_fma(f, qt2, qa2), sc, qb2);
        mu = j == 0 ? fmaxf(u2[0][0], u2[0][1]) : fmaxf(mu, fmaxf(u2[j][0], u2[j][1]));
    };
    auto decide = [&](Pending& R) __attribute__((always_inline)) {
        if (!R.have) return;
        R.have = false;
        float smax = R.sr[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) smax = fmaxf(smax, R.sr[r >> 2][r & 3]);
        const unsigned row0 = (unsigned)R.sb * 32u + 4u * (unsigned)lh;
        const bool whole = (long long)R.sb * 32 + 32 <= n_rows;  // wave-uniform
        if (whole && (MODE != 1 || __ballot(q_valid && mu >= known) == 0ull)) {
            const float cl = mu - 2.f * fmaf(qa, smax, qb);
            if (q_valid && cl > lmax) {  // a new record for this lane: remember the row
                lmax = cl;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (u2[r >> 1][r & 1] == mu) lrow = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float up = u2[r >> 1][r & 1];
                const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
                const bool ok = (long long)row < n_rows && q_valid;
                const float lr_ = up - 2.f * fmaf(qa, R.sr[r >> 2][r & 3], qb);
                if (ok && lr_ > lmax) { lmax = lr_; lrow = row; }
                if (MODE == 1) {
                    const bool hit = ok && up >= known;
                    const u64 m = __ballot(hit);
                    if (m) {  // wave-uniform
                        const int slot_i = my_cnt + __popcll(m & lt);
                        if (hit && slot_i < kWaveCandCap) my_list[slot_i] = pack_cand8(q_base + qlocal, up, row);
                        my_cnt += __popcll(m);
                    }
                }
            }
        }
        if (MODE == 1) known = fmaxf(known, fmaxf(lmax, __shfl_xor(lmax, 32)));  // the two lanes of a query share their bounds
    };
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_t = 0;
    auto stamp = [&](int seg) __attribute__((always_inline)) {
        if (MDR_I8_ABL != 9 || MODE != 1) return;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_readcyclecounter();
        if (seg >= 0) st_sum[seg] += now - st_t;
        st_t = now;
        __builtin_amdgcn_sched_barrier(0);
    };
    stamp(-1);
    for (int it = 0; it < n_it; ++it) {
        if (it + NS - 2 < n_it) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((CPW + SPS) * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW * (NS - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(0);
        if (MODE == 1 && ((it + b) & 31) == 31 && wave_active) {
            float hm = fmaxf(lmax, __shfl_xor(lmax, 32));
            float kn = known;
            if (lane < 32 && q_valid) {
                if (hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
                unsigned g = load_u32_l2(gmax + qlocal);
                if (g) kn = fmaxf(kn, unord32(g));
            }
            known = __shfl(kn, l31);
        }
        if (it + NS - 1 < n_it && MDR_I8_ABL != 6) issue_stage(b + (it + NS - 1) * G, lds + ((it + NS - 1) % NS) * ST_BYTES);
        stamp(1);
        if (!wave_active || MDR_I8_ABL == 5) continue;

        static_for<0, SPS>([&](auto hc) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            Pending& Wp = P[h];      // this super-block's accumulator set
            Pending& Rp = P[h ^ 1];  // the one whose epilogue is still pending: the super-block before this one
            const int sb_idx = SPS * (b + it * G) + h;
            if (sb_idx >= n_sb) return;  // wave-uniform: the corpus ends inside this stage
            const char* slot = lds + (it % NS) * ST_BYTES + h * SB_BYTES;
            // this lane's 16 row scales: rows 8 j + 4 lh .. + 3, j = 0..3
#pragma unroll
            for (int j = 0; j < 4; ++j) Wp.sr[j] = *(const f32x4*)(slot + 2 * NKB8 * kFragBytes + (8 * j + 4 * lh) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Wp.sr[0]), "+v"(Wp.sr[1]), "+v"(Wp.sr[2]), "+v"(Wp.sr[3]));
            if (MDR_I8_ABL == 3) {
#pragma unroll
                for (int e = 0; e < 16; ++e) Wp.acc[e] = qf[e][0] + it;
            } else {
                Wp.acc = mfma_chain8x32<NKB8>(slot + rd_off, qf, [&](auto slc) __attribute__((always_inline)) {
                    constexpr int sl = decltype(slc)::value;
                    if constexpr (MDR_I8_ABL != 2) {
                        if constexpr (sl < 8) bounds(Rp, slc);
                        else if constexpr (sl == 8) decide(Rp);
                    }
                });
            }
            Wp.sb = sb_idx;
            Wp.have = true;
        });
        stamp(2);
    }
    if (wave_active && MDR_I8_ABL != 2 && MDR_I8_ABL != 5) {  // the last super-block's epilogue (at most one is pending)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (P[h].have) {
                static_for<0, 8>([&](auto jc) __attribute__((always_inline)) { bounds(P[h], jc); });
                decide(P[h]);
            }
    }
#if MDR_I8_ABL == 9
    if (MODE == 1 && threadIdx.x == 0) {
#pragma unroll
        for (int e = 0; e < 5; ++e) atomicAdd(&g_i8_stamp[e], st_sum[e]);
        atomicAdd(&g_i8_stamp[7], (unsigned long long)n_it);
    }
#endif
    {
        const float hm = fmaxf(lmax, __shfl_xor(lmax, 32));
        if (lane < 32 && q_valid && hm > -FLT_MAX) atomicMax(gmax + qlocal, ord32(hm));
        if (q_valid && lmax == hm && hm > -FLT_MAX) atomicMax(gstar + qlocal, ((u64)ord32(lmax) << 32) | lrow);
    }
    if (MODE == 1 && lane == 0) {
        cand_cnt[b * 8 + wave] = my_cnt < kWaveCandCap ? my_cnt : kWaveCandCap;
        if (my_cnt > kWaveCandCap) *overflow = 1;
        if (my_cnt) atomicAdd(overflow + 3, my_cnt);  // ctl8[3]: candidates emitted by this pass (the refinement's guard)
    }
}

// exact re-scoring of the int8 tier's candidates: as mips_refine_kernel, after dropping every candidate whose (rounded-up) upper
// bound lies below the FINAL largest lower bound of its query -- most of a no-clear-winner query's candidates were emitted early,
// against a `known` that the pass later raised. ctl8[1] counts the candidates that are really re-scored.
// The row with the best lower bound of every query is re-scored FIRST: its exact score seeds the thresholds of mips_refine8_kernel,
// which then only gathers the rows whose upper bound reaches an exact score (a handful per query instead of hundreds).
__global__ void __launch_bounds__(256)
mips_star8_kernel(const char* __restrict__ Xhi, const char* __restrict__ Xlo, int nkb, const float* __restrict__ q, const u64* __restrict__ gstar, int nq,
                  u64* __restrict__ best, float xs) {
    const int qi = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (qi >= nq) return;
    const u64 key = gstar[qi];
    if (key == 0) return;
    const unsigned row = (unsigned)key;
    const float acc = exact_dot16<false>(Xhi, Xlo, nkb, q + (size_t)qi * (nkb * 32), row, sub, xs);
    if (sub == 0) atomicMax(best + qi, make_key(acc, row));
}

// `limit`: emitted candidates (ctl8[3], summed by the main pass) beyond which filtering and re-scoring them would cost more than the
// fp16 screen pass behind this tier (data for which the int8 bound is loose: rows with a large common mean, all-ties corpora):
// the tier then declares itself overflowed.
__global__ void __launch_bounds__(256)
mips_refine8_kernel(const char* __restrict__ Xhi, const char* __restrict__ Xlo, int nkb, const float* __restrict__ q, const u64* __restrict__ cand,
                    const int* __restrict__ cand_cnt, const unsigned* __restrict__ gmax, u64* __restrict__ best, int* __restrict__ ctl8, int limit,
                    const f32x4* __restrict__ qab /* [3] = q.c + slack: exact score -> centred units (the candidates' bounds are centred) */, float xs) {
    if (ctl8[0] || ctl8[3] > limit) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl8[0] = 1;
        return;
    }
    const int n = cand_cnt[blockIdx.x];
    if (n == 0) return;
    const u64* list = cand + (size_t)blockIdx.x * kWaveCandCap;
    const int sub = threadIdx.x & 15;
    const int d = nkb * 32;
    // A list belongs to ONE wave of the screen kernel: its candidates share at most 32 consecutive queries (aligned to 32). thr[] is
    // this block's copy of "an exact score somebody already reached for that query" (top 16 bits of the ordered score): seeded from
    // `best` once, raised by this block's own re-scorings. A row whose upper bound lies below it cannot win (an equal score survives,
    // so the lowest id still wins ties). (Reading `best` itself per candidate -- 4e5 uncached loads of 200 hot words -- doubled the
    // kernel's time.)
    __shared__ unsigned thr[32];
    __shared__ float qoff[32];
    const unsigned qb32 = (unsigned)(list[0] >> 48) & ~31u;
    if (threadIdx.x < 32) {
        const u64 kb = best[qb32 + threadIdx.x];
        const float off = qab[qb32 + threadIdx.x][3];
        qoff[threadIdx.x] = off;
        thr[threadIdx.x] = kb ? ord32(key_score(kb) - off) >> 16 : 0u;  // (truncation rounds the threshold DOWN: safe)
    }
    __syncthreads();
    int kept = 0;
    for (int c = threadIdx.x >> 4; c < n; c += 16) {
        const u64 e = list[c];
        const unsigned qi = (unsigned)(e >> 48), u16 = (unsigned)(e >> 32) & 0xFFFFu, row = (unsigned)e;
        if (u16 < (gmax[qi] >> 16)) continue;  // U < final max L: cannot be the best row
        if (u16 < thr[qi & 31]) continue;
        const float acc = exact_dot16<false>(Xhi, Xlo, nkb, q + (size_t)qi * d, row, sub, xs);
        if (sub == 0) {
            atomicMax(best + qi, make_key(acc, row));
            atomicMax(&thr[qi & 31], ord32(acc - qoff[qi & 31]) >> 16);
            ++kept;
        }
    }
    if (sub == 0 && kept) atomicAdd(ctl8 + 1, kept);
}

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

}  // namespace
}  // namespace mdr

// ======================================================================================================
// host side
// ======================================================================================================
using namespace mdr;

struct mdr_index {
    int d = 0, nkb = 0, storage = 0, device = 0;
    int num_cus = 256;
    long long ntotal = 0;
    long long cap_rows = 0;  // multiple of 32 (one screen-kernel super-block)
    char* hi = nullptr;      // fragment-tiled planes, cap_rows * d * 2 bytes each
    char* lo = nullptr;
    char* i8 = nullptr;      // int8 screening plane (F32X2H storage, d = 768): cap_rows / 32 super-blocks of i8_sb_bytes(d / 64), see mips_screen8_kernel
    float* centre = nullptr; // [3 d] c | 1/w | w: centre and per-column scale of the int8 plane (col_sum_kernel), (0, 1, 1) until the first add() sets
                             // them, frozen afterwards; [3 d .. 5 d) scratch for the column sums
    bool centre_set = false;
    int xexp = 0;            // F32X2H: the planes hold x * 2^-xexp (see convert_to_frag_kernel); fitted to the data by add(), grown by a rescale
    bool xexp_set = false;
    int* flags = nullptr;    // device ints: [0] range error seen by add(), [1] same for queries (ignored), [2] max row |x|^2 (float bits),
                             // [8], [9] int8 tier: max row scale, max s_r (L1(x8_r)/2 + d/4) (float bits)
    void* stage = nullptr;   // device staging for host-sourced add()
    size_t stage_bytes = 0;
    int variant = 0;
    const char* last_kernel = "none";
};

namespace {


size_t plane_bytes_per_row(const mdr_index* h) { return (size_t)h->d * 2; }
float row_unscale(const mdr_index* h) { return ldexpf(1.f, h->xexp); }   // 2^E: stored rows -> the caller's scale
float row_scale_inv(const mdr_index* h) { return ldexpf(1.f, -h->xexp); }  // 2^-E
long long pad32(long long n) { return (n + 31) / 32 * 32; }

// the int8 screening plane exists for the storage / dimension the k = 1 screen path serves; MDR_MIPS_I8=0 (read at index creation) leaves it out
bool wants_i8(const mdr_index* h) {
    static const bool off = getenv("MDR_MIPS_I8") && atoi(getenv("MDR_MIPS_I8")) == 0;
    return !off && h->storage != MDR_STORE_BF16 && h->d == 768;
}

int grow(mdr_index* h, long long need_rows, hipStream_t st) {
    long long need = pad32(need_rows);
    if (need <= h->cap_rows) return MDR_OK;
    long long ncap = need;  // first reservation is exact; later ones grow by 1.5x
    if (h->cap_rows) {
        long long geo = pad32(h->cap_rows + h->cap_rows / 2);
        if (geo > ncap) ncap = geo;
    }
    const size_t nbytes = (size_t)ncap * plane_bytes_per_row(h);
    const size_t used = (size_t)pad32(h->ntotal) * plane_bytes_per_row(h);
    char* planes[2] = {nullptr, nullptr};
    char* old[2] = {h->hi, h->lo};
    const int nplanes = h->storage == MDR_STORE_BF16 ? 1 : 2;
    for (int i = 0; i < nplanes; ++i) {
        MDR_HIP_TRY(hipMalloc((void**)&planes[i], nbytes));
        if (used) MDR_HIP_TRY(hipMemcpyAsync(planes[i], old[i], used, hipMemcpyDeviceToDevice, st));
        MDR_HIP_TRY(hipMemsetAsync(planes[i] + used, 0, nbytes - used, st));
    }
    char* n8 = nullptr;
    if (wants_i8(h) && !h->centre) {
        MDR_HIP_TRY(hipMalloc((void**)&h->centre, (size_t)h->d * 5 * 4));
        float init[3 * 1024];  // d <= 1024: c = 0, 1/w = 1, w = 1 until the first add() measures them
        for (int i = 0; i < 3 * h->d; ++i) init[i] = i < h->d ? 0.f : 1.f;
        MDR_HIP_TRY(hipMemcpy(h->centre, init, (size_t)h->d * 3 * 4, hipMemcpyHostToDevice));
    }
    if (wants_i8(h)) {
        const size_t sbb = i8_sb_bytes(h->d / 64), nb8 = (size_t)(ncap / 32 + 1) * sbb, used8 = (size_t)(pad32(h->ntotal) / 32) * sbb;  // + 1: the wide kernel's stages are super-block pairs
        MDR_HIP_TRY(hipMalloc((void**)&n8, nb8));
        if (used8) MDR_HIP_TRY(hipMemcpyAsync(n8, h->i8, used8, hipMemcpyDeviceToDevice, st));
        MDR_HIP_TRY(hipMemsetAsync(n8 + used8, 0, nb8 - used8, st));
    }
    MDR_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < nplanes; ++i)
        if (old[i]) MDR_HIP_TRY(hipFree(old[i]));
    if (h->i8) MDR_HIP_TRY(hipFree(h->i8));
    h->hi = planes[0];
    h->lo = planes[1];
    h->i8 = n8;
    h->cap_rows = ncap;
    return MDR_OK;
}

template <typename T>
int launch_convert(bool bf, const T* src_dev, long long n_valid, long long n_total, int d, long long row0, char* dst_hi, char* dst_lo, int* flags,
                   float xinv, hipStream_t st) {
    long long threads = n_total * (d / 8);
    if (threads == 0) return MDR_OK;
    long long blocks = (threads + 255) / 256;
    if (bf)
        hipLaunchKernelGGL((convert_to_frag_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, src_dev, n_valid, n_total, d, row0, dst_hi, dst_lo, flags, 1.0f);
    else
        hipLaunchKernelGGL((convert_to_frag_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, src_dev, n_valid, n_total, d, row0, dst_hi, dst_lo, flags, xinv);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

constexpr long long kCentreRows = 65536;  // rows of the first add() the int8 plane's centre is averaged over

template <typename T>
int launch_add(mdr_index* h, const T* src_dev, long long n, long long row0, hipStream_t st) {
    const float xinv = h->storage == MDR_STORE_BF16 ? 1.0f : row_scale_inv(h);
    int rc = launch_convert(h->storage == MDR_STORE_BF16, src_dev, n, n, h->d, row0, h->hi, h->lo, h->flags, xinv, st);
    if (rc) return rc;
    hipLaunchKernelGGL(row_norm2_max_kernel<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, src_dev, n, h->d, h->flags, xinv);
    if (h->i8 && n > 0) {
        if (!h->centre_set) {  // the first rows this index ever sees define the centre of its int8 plane (any fixed vector is correct)
            const long long nc = n < kCentreRows ? n : kCentreRows;
            float* sums = h->centre + 3 * (size_t)h->d;
            MDR_HIP_TRY(hipMemsetAsync(sums, 0, (size_t)h->d * 8, st));
            const unsigned gy = (unsigned)((nc + 3) / 4 < 256 ? (nc + 3) / 4 : 256);
            hipLaunchKernelGGL(col_sum_kernel<T>, dim3((unsigned)((h->d + 63) / 64), gy), dim3(256), 0, st, src_dev, nc, h->d, sums);
            hipLaunchKernelGGL(centre_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)sums, h->d, 1.0f / (float)nc, h->centre);
            h->centre_set = true;
        }
        hipLaunchKernelGGL(convert_to_i8_kernel<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, src_dev, n, h->d, row0, h->i8, h->flags + 8,
                           (const float*)h->centre);
    }
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

__global__ void scale_flag_float_kernel(int* __restrict__ flag, float f) { *flag = __float_as_int(__int_as_float(*flag) * f); }

// Fit the index exponent E (planes hold x * 2^-E) to the rows about to be converted; rows [0, row0) are already stored. Synchronises.
template <typename T>
int fit_exponent(mdr_index* h, const T* src_dev, long long n, long long row0, hipStream_t st) {
    if (h->storage == MDR_STORE_BF16 || n == 0) return MDR_OK;
    MDR_HIP_TRY(hipMemsetAsync(h->flags + 4, 0, sizeof(int), st));
    const long long count = n * (long long)h->d;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_kernel<T>, dim3(blocks), dim3(256), 0, st, src_dev, count, h->flags);
    MDR_HIP_TRY(hipGetLastError());
    int bits = 0;
    MDR_HIP_TRY(hipMemcpyAsync(&bits, h->flags + 4, sizeof(int), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    float m;
    memcpy(&m, &bits, 4);
    if (!(m <= 3.0e38f)) return set_error(MDR_E_RANGE, "add(): a value is non-finite; rows were not added");
    if (m == 0.f) return MDR_OK;
    int e = 0;
    (void)frexpf(m, &e);           // m in [2^(e-1), 2^e)
    const int target = e - 10;     // m * 2^-target in [2^9, 2^10): headroom of 2^5 before fp16 overflows
    if (!h->xexp_set) {
        h->xexp = target;
        h->xexp_set = true;
        return MDR_OK;
    }
    if (e - h->xexp > 15) {        // m * 2^-E >= 2^15 would not fit: grow E and shrink what is stored by the same power of two (exact)
        const float f = ldexpf(1.f, h->xexp - target);
        const long long n_vec8 = (row0 + 15) / 16 * 16 * (long long)h->d / 8;
        if (n_vec8 > 0) {
            hipLaunchKernelGGL(rescale_planes_kernel, dim3((unsigned)((n_vec8 + 255) / 256)), dim3(256), 0, st, h->hi, h->lo, n_vec8, f);
            hipLaunchKernelGGL(scale_flag_float_kernel, dim3(1), dim3(1), 0, st, h->flags + 2, f * f);  // max row |x|^2 in stored units
            MDR_HIP_TRY(hipGetLastError());
        }
        h->xexp = target;
    }
    return MDR_OK;
}

int add_any(mdr_index* h, const void* src_dev, int dtype, long long n, long long row0, hipStream_t st) {
    int rc = MDR_OK;
    switch (dtype) {  // (before anything is written: a non-finite value rejects the rows here)
        case MDR_DT_F32: rc = fit_exponent(h, (const float*)src_dev, n, row0, st); break;
        case MDR_DT_BF16: rc = fit_exponent(h, (const unsigned short*)src_dev, n, row0, st); break;
        case MDR_DT_F16: rc = fit_exponent(h, (const _Float16*)src_dev, n, row0, st); break;
        default: break;
    }
    if (rc) return rc;
    switch (dtype) {
        case MDR_DT_F32: return launch_add(h, (const float*)src_dev, n, row0, st);
        case MDR_DT_BF16: return launch_add(h, (const unsigned short*)src_dev, n, row0, st);
        case MDR_DT_F16: return launch_add(h, (const _Float16*)src_dev, n, row0, st);
        default: return set_error(MDR_E_INVALID, "unknown src_dtype %d", dtype);
    }
}

size_t elem_size(int dtype) { return dtype == MDR_DT_F32 ? 4 : 2; }

bool is_bf16(const mdr_index* h) { return h->storage == MDR_STORE_BF16; }
// MDR_MIPS_WIDE=0 keeps every call on the 128-queries-per-pass kernels (measurement knob)
bool wide_pass(int nq) {
    static const bool off = getenv("MDR_MIPS_WIDE") && atoi(getenv("MDR_MIPS_WIDE")) == 0;
    return !off && nq > kStreamQ;
}
bool stream_kernel_supports(const mdr_index* h, int k) { return !is_bf16(h) && h->d == 768 && k <= 128; }
bool screen_kernel_supports(const mdr_index* h, int k) { return h->d == 768 && k <= 128; }

enum Path { PATH_GENERIC = 1, PATH_STREAM = 2, PATH_SCREEN = 3 };

struct SearchPlan {
    int path;
    int G;    // workgroups of the stream / screen kernels
    int Gx;   // workgroups of the exact stream kernel (== G on the stream path; the fallback behind the screen kernels)
    int Gg;   // workgroups of the generic kernel (when its lists are needed)
    bool lists_stream, lists_generic;
    size_t off_qhi, off_qlo, off_bound, off_qscale, off_best, off_gmax, off_scand, off_sctl, off_cand, off_cnt, off_kth, total;
    bool i8;  // the int8 screening tier runs in front of the fp16 screen (k == 1)
    int G8w;  // workgroups of its 32-queries-per-wave kernel
    int G8;   // its workgroups: TWO per CU (24.25 KiB super-blocks: three slots are 73 KiB), one's barrier and epilogue under the other's MFMAs.
              // Measured at 5 M rows, planted queries, whole call: 1 per CU 1.013 ms, 1 per CU with 64-row stages 0.995 ms, 2 per CU 0.907 ms.
    size_t off_q8, off_qab, off_ctl8, off_gstar;
};

#ifndef MDR_I8W_SLOTS
#define MDR_I8W_SLOTS 3  // the 32-queries-per-wave int8 kernel needs > 128 VGPRs: ONE workgroup per CU; its stages are 64 rows (48.5 KiB), two in flight
#endif
#ifndef MDR_I8_SLOTS
#define MDR_I8_SLOTS 3  // variant-build knob: LDS ring depth of the int8 screen kernels (3: two workgroups per CU, 4-6: one)
#endif
// variant 4 = the screen path WITHOUT the int8 tier (tests and A/B runs)
// (run_screen8 serves ONE group of at most kStreamQ queries; more than that goes to the 32-queries-per-wave kernel, which loops over
// groups of 256 -- or, with MDR_MIPS_WIDE=0, stays on the fp16 screen, which loops over groups of 128)
bool i8_tier(const mdr_index* h, int path, int nq, int k) {
    return h->i8 != nullptr && h->variant != 4 && path == PATH_SCREEN && k == 1 && nq < 65536 && (nq <= kStreamQ || wide_pass(nq));
}

SearchPlan make_plan(const mdr_index* h, int nq, int k) {
    SearchPlan p{};
    const int v = h->variant;
    if (v == PATH_GENERIC) p.path = PATH_GENERIC;
    else if (v == PATH_STREAM) p.path = stream_kernel_supports(h, k) ? PATH_STREAM : PATH_GENERIC;
    else if (screen_kernel_supports(h, k)) p.path = PATH_SCREEN;  // auto or forced screen
    else if (stream_kernel_supports(h, k)) p.path = PATH_STREAM;
    else p.path = PATH_GENERIC;
    const long long n_rb = (h->ntotal + 15) / 16;
    const long long units = p.path == PATH_SCREEN ? (h->ntotal + 31) / 32 : n_rb;
    p.G = (int)(units < h->num_cus ? (units > 0 ? units : 1) : h->num_cus);
    if (p.G > 1024) p.G = 1024;  // kth_of_maxima_kernel holds one value per workgroup in LDS
    p.Gx = (int)(n_rb < h->num_cus ? (n_rb > 0 ? n_rb : 1) : h->num_cus);
    const long long gg = (long long)h->num_cus * 2;
    p.Gg = (int)(n_rb < gg ? (n_rb > 0 ? n_rb : 1) : gg);
    // which candidate-list workspaces this call can touch (incl. the conditional exact pass behind the screen kernel)
    p.lists_stream = (p.path == PATH_STREAM || (p.path == PATH_SCREEN && !is_bf16(h))) && k > 1;
    p.lists_generic = p.path == PATH_GENERIC || (p.path == PATH_SCREEN && is_bf16(h));
    const bool screenk = p.path == PATH_SCREEN && k > 1;
    const bool frag = p.path != PATH_GENERIC;
    const size_t nq_pad = (size_t)((nq + kWideQ - 1) / kWideQ) * kWideQ;  // covers both group sizes (128 and 256 queries per pass)
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    p.off_qhi = take(frag ? nq_pad * h->d * 2 : 0);
    p.off_qlo = take(frag && !is_bf16(h) ? nq_pad * h->d * 2 : 0);
    p.off_bound = take(frag ? nq_pad * 4 : 0);
    p.off_qscale = take(frag ? nq_pad * 4 : 0);
    p.off_best = take((size_t)(nq > 0 ? nq : 1) * 8);
    p.off_gmax = take(p.path == PATH_SCREEN ? nq_pad * 4 : 0);
    // k == 1: one private list per wave; k > 1: the [G][kStreamQ] sample maxima
    p.i8 = i8_tier(h, p.path, nq, k);
    {
        const long long per_cu = MDR_I8_SLOTS <= 3 ? 2 : 1;  // workgroups of the int8 kernels per CU (LDS: 3 slots are 73 KiB, 6 are 146 KiB)
        p.G8 = (int)(units < per_cu * h->num_cus ? (units > 0 ? units : 1) : per_cu * h->num_cus);
        p.G8w = (int)((units + 1) / 2 < h->num_cus ? ((units + 1) / 2 > 0 ? (units + 1) / 2 : 1) : h->num_cus);  // the 32-queries-per-wave kernel: one per CU, stages of two super-blocks
    }
    const size_t gl = p.i8 && p.G8 > p.G ? (size_t)p.G8 : (size_t)p.G;  // workgroups that own candidate lists
    p.off_scand = take(p.path != PATH_SCREEN ? 0 : (k == 1 ? gl * 8 * kWaveCandCap * 8 : (size_t)p.G * kWideQ * 4));
    p.off_sctl = take(p.path == PATH_SCREEN ? 256 + gl * 8 * 4 : 0);            // [0] overflow flag, [64..] per-wave counts
    // the screen-k lists and the lists of its conditional exact pass (which runs after them in stream order) share one region
    size_t lists = p.lists_stream ? (size_t)p.Gx * kStreamQ * kStreamCap : (p.lists_generic ? (size_t)p.Gg * kGenericQ * kGenericCap : 0);
    size_t slots = p.lists_stream ? (size_t)p.Gx * kStreamQ : (p.lists_generic ? (size_t)p.Gg * kGenericQ : 0);
    if (screenk) {
        const size_t qc = wide_pass(nq) ? kWideQ : kStreamQ;
        const size_t l2 = (size_t)p.G * qc * kScreenKCap, s2 = (size_t)p.G * qc;
        lists = lists > l2 ? lists : l2;
        slots = slots > s2 ? slots : s2;
    }
    p.off_cand = take(lists * 8);
    p.off_cnt = take(slots * 4);
    p.off_kth = take(slots * 8);
    p.off_q8 = take(p.i8 ? nq_pad * h->d : 0);
    p.off_qab = take(p.i8 ? nq_pad * 16 : 0);
    p.off_ctl8 = take(p.i8 ? 256 : 0);  // [0] a candidate list of the int8 tier overflowed -> the fp16 screen runs
    p.off_gstar = take(p.i8 ? nq_pad * 8 : 0);
    p.total = o + 256;
    return p;
}

// generic kernel + merge over all queries in groups of kGenericQ; every launch is skipped on the device when *run_if == 0
template <bool BF>
int run_generic(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, int k, float* D_dev, long long* I_dev, long long id_offset,
                const int* run_if, hipStream_t st) {
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    u64* kth = (u64*)(ws + p.off_kth);
    const int n_rb = (int)((h->ntotal + 15) / 16);
    for (int q0 = 0; q0 < nq; q0 += kGenericQ) {
        const int nqg = nq - q0 < kGenericQ ? nq - q0 : kGenericQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.Gg * kGenericQ * 4, st));
        MDR_HIP_TRY(hipMemsetAsync(kth, 0, (size_t)p.Gg * kGenericQ * 8, st));
        hipLaunchKernelGGL((mips_generic_kernel<BF>), dim3(p.Gg), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, (long long)h->ntotal, n_rb, h->nkb,
                           q_dev + (size_t)q0 * h->d, nqg, cand, cnt, kth, k, run_if, row_unscale(h));
        hipLaunchKernelGGL(merge_lists_kernel, dim3(nqg), dim3(256), 0, st, (const u64*)cand, (const int*)cnt, (const u64*)kth, p.Gg, kGenericQ, kGenericCap,
                           k, D_dev + (size_t)q0 * k, I_dev + (size_t)q0 * k, id_offset, run_if, (const float*)nullptr);
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

template <bool BF>
int run_screen(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, const char* qhi, u64* best, hipStream_t st,
               const int* run_if = nullptr) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 0, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 1, BF>, (int)lds_bytes);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    unsigned* gmax = (unsigned*)(ws + p.off_gmax);
    u64* scand = (u64*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    int* wave_cnt = sctl + 64;
    const int ngroups = (nq + kStreamQ - 1) / kStreamQ;
    const int nq_pad = ngroups * kStreamQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const size_t qgroup_bytes = (size_t)kStreamQ * h->d * 2;
    MDR_HIP_TRY(hipMemsetAsync(gmax, 0, (size_t)nq_pad * 4, st));
    MDR_HIP_TRY(hipMemsetAsync(sctl, 0, 256, st));
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
        const char* qg = qhi + gi * qgroup_bytes;
        hipLaunchKernelGGL((mips_screen_kernel<NKB, 0, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gi * kStreamQ), nqg, gi * kStreamQ, gmax + (size_t)gi * kStreamQ, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_screen_kernel<NKB, 1, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gi * kStreamQ), nqg, gi * kStreamQ, gmax + (size_t)gi * kStreamQ, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_refine_kernel<BF>), dim3(p.G * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev,
                           (const u64*)scand, (const int*)wave_cnt, best, row_unscale(h), run_if);
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// k == 1, at most 128 queries, int8 plane present: the int8 tier (sample pass, main pass, exact re-scoring of its candidates)
int run_screen8(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, u64* best, hipStream_t st) {
    constexpr int NKB8 = 12, NS = MDR_I8_SLOTS;
    const size_t lds_bytes = NS * (size_t)(2 * NKB8 * kFragBytes + kI8Tail);
    int rc_ = ensure_dynamic_lds((const void*)mips_screen8_kernel<NKB8, 0, NS>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen8_kernel<NKB8, 1, NS>, (int)lds_bytes);
    if (rc_) return rc_;
    unsigned* gmax = (unsigned*)(ws + p.off_gmax);
    u64* scand = (u64*)(ws + p.off_scand);
    int* wave_cnt = (int*)(ws + p.off_sctl) + 64;
    int* ctl8 = (int*)(ws + p.off_ctl8);
    char* q8 = ws + p.off_q8;
    f32x4* qab = (f32x4*)(ws + p.off_qab);
    const int nq_pad = kStreamQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    u64* gstar = (u64*)(ws + p.off_gstar);
    MDR_HIP_TRY(hipMemsetAsync(gmax, 0, (size_t)nq_pad * 4, st));
    MDR_HIP_TRY(hipMemsetAsync(gstar, 0, (size_t)nq_pad * 8, st));
    MDR_HIP_TRY(hipMemsetAsync(ctl8, 0, 256, st));
    hipLaunchKernelGGL(prep_queries_i8_kernel, dim3((nq_pad + 3) / 4), dim3(256), 0, st, q_dev, nq, nq_pad, h->d, (const int*)(h->flags + 8), q8, qab,
                       (const float*)h->centre);
    hipLaunchKernelGGL((mips_screen8_kernel<NKB8, 0, NS>), dim3(p.G8), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, (const char*)q8,
                       (const f32x4*)qab, nq, 0, gmax, scand, wave_cnt, ctl8, gstar, (const u64*)best);
    // the sample pass's best-lower-bound rows, re-scored exactly: a first `known` that is up to 2 B tighter than their lower bounds
    hipLaunchKernelGGL(mips_star8_kernel, dim3((nq + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)gstar, nq, best, row_unscale(h));
    hipLaunchKernelGGL((mips_screen8_kernel<NKB8, 1, NS>), dim3(p.G8), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, (const char*)q8,
                       (const f32x4*)qab, nq, 0, gmax, scand, wave_cnt, ctl8, gstar, (const u64*)best);
    hipLaunchKernelGGL(mips_star8_kernel, dim3((nq + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)gstar, nq, best, row_unscale(h));
    hipLaunchKernelGGL(mips_refine8_kernel, dim3(p.G8 * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)scand,
                       (const int*)wave_cnt, (const unsigned*)gmax, best, ctl8, kI8RefinePerQuery * nq, (const f32x4*)qab, row_unscale(h));
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

// k == 1, more than 128 queries: the same three launches per group of 256 queries on the 32-queries-per-wave kernel
template <bool BF>
int run_screen32(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, const char* qhi, u64* best, hipStream_t st,
                 const int* run_if = nullptr) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen32_kernel<NKB, 0, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen32_kernel<NKB, 1, BF>, (int)lds_bytes);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    unsigned* gmax = (unsigned*)(ws + p.off_gmax);
    u64* scand = (u64*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    int* wave_cnt = sctl + 64;
    const int ngroups = (nq + kWideQ - 1) / kWideQ;
    const int nq_pad = ngroups * kWideQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const size_t qgroup_bytes = (size_t)kWideQ * h->d * 2;
    MDR_HIP_TRY(hipMemsetAsync(gmax, 0, (size_t)nq_pad * 4, st));
    MDR_HIP_TRY(hipMemsetAsync(sctl, 0, 256, st));
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kWideQ < kWideQ ? nq - gi * kWideQ : kWideQ;
        const char* qg = qhi + gi * qgroup_bytes;
        hipLaunchKernelGGL((mips_screen32_kernel<NKB, 0, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gi * kWideQ), nqg, gi * kWideQ, gmax + (size_t)gi * kWideQ, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_screen32_kernel<NKB, 1, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gi * kWideQ), nqg, gi * kWideQ, gmax + (size_t)gi * kWideQ, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_refine_kernel<BF>), dim3(p.G * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev,
                           (const u64*)scand, (const int*)wave_cnt, best, row_unscale(h), run_if);
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// k == 1, more than 128 queries, int8 plane present: the int8 tier per group of 256 queries
int run_screen8w(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, u64* best, hipStream_t st) {
    constexpr int NKB8 = 12, NS = MDR_I8W_SLOTS;
    const size_t lds_bytes = NS * 2 * (size_t)(2 * NKB8 * kFragBytes + kI8Tail);  // a stage of this kernel = two super-blocks
    int rc_ = ensure_dynamic_lds((const void*)mips_screen8w_kernel<NKB8, 0, NS>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen8w_kernel<NKB8, 1, NS>, (int)lds_bytes);
    if (rc_) return rc_;
    unsigned* gmax = (unsigned*)(ws + p.off_gmax);
    u64* scand = (u64*)(ws + p.off_scand);
    int* wave_cnt = (int*)(ws + p.off_sctl) + 64;
    int* ctl8 = (int*)(ws + p.off_ctl8);
    char* q8 = ws + p.off_q8;
    f32x4* qab = (f32x4*)(ws + p.off_qab);
    const int ngroups = (nq + kWideQ - 1) / kWideQ;
    const int nq_pad = ngroups * kWideQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    u64* gstar = (u64*)(ws + p.off_gstar);
    MDR_HIP_TRY(hipMemsetAsync(gmax, 0, (size_t)nq_pad * 4, st));
    MDR_HIP_TRY(hipMemsetAsync(gstar, 0, (size_t)nq_pad * 8, st));
    MDR_HIP_TRY(hipMemsetAsync(ctl8, 0, 256, st));
    hipLaunchKernelGGL(prep_queries_i8_kernel, dim3((nq_pad + 3) / 4), dim3(256), 0, st, q_dev, nq, nq_pad, h->d, (const int*)(h->flags + 8), q8, qab,
                       (const float*)h->centre);
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kWideQ < kWideQ ? nq - gi * kWideQ : kWideQ;
        const char* qg = q8 + (size_t)gi * kWideQ * h->d;
        if (gi) MDR_HIP_TRY(hipMemsetAsync(ctl8 + 3, 0, sizeof(int), st));  // emitted-candidate total of this group's pass
        hipLaunchKernelGGL((mips_screen8w_kernel<NKB8, 0, NS>), dim3(p.G8w), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, qg,
                           (const f32x4*)(qab + (size_t)gi * kWideQ), nqg, gi * kWideQ, gmax + (size_t)gi * kWideQ, scand, wave_cnt, ctl8, gstar + (size_t)gi * kWideQ,
                           (const u64*)best);
        hipLaunchKernelGGL(mips_star8_kernel, dim3((nqg + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb,
                           q_dev + (size_t)gi * kWideQ * h->d, (const u64*)(gstar + (size_t)gi * kWideQ), nqg, best + (size_t)gi * kWideQ, row_unscale(h));
        hipLaunchKernelGGL((mips_screen8w_kernel<NKB8, 1, NS>), dim3(p.G8w), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, qg,
                           (const f32x4*)(qab + (size_t)gi * kWideQ), nqg, gi * kWideQ, gmax + (size_t)gi * kWideQ, scand, wave_cnt, ctl8, gstar + (size_t)gi * kWideQ,
                           (const u64*)best);
        hipLaunchKernelGGL(mips_star8_kernel, dim3((nqg + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb,
                           q_dev + (size_t)gi * kWideQ * h->d, (const u64*)(gstar + (size_t)gi * kWideQ), nqg, best + (size_t)gi * kWideQ, row_unscale(h));
        hipLaunchKernelGGL(mips_refine8_kernel, dim3(p.G8w * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)scand,
                           (const int*)wave_cnt, (const unsigned*)gmax, best, ctl8, kI8RefinePerQuery * nqg, (const f32x4*)qab, row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// 2 <= k <= 128: sample pass -> k-th of the workgroup maxima -> screen-k kernel -> merge/refine, per query group;
// sctl[0] = overflow flag for the conditional exact pass
template <bool BF>
int run_screenk(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, int k, const char* qhi, float* D_dev, long long* I_dev,
                long long id_offset, hipStream_t st) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    const size_t merge_lds = (size_t)kMergeKLds * 8;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 2, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screenk_kernel<NKB, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)merge_screenk_kernel<BF>, (int)merge_lds);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    float* tau0 = (float*)(ws + p.off_gmax);
    unsigned* wgmax = (unsigned*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    const int ngroups = (nq + kStreamQ - 1) / kStreamQ;
    const int nq_pad = ngroups * kStreamQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const size_t qgroup_bytes = (size_t)kStreamQ * h->d * 2;
    MDR_HIP_TRY(hipMemsetAsync(sctl, 0, 256, st));
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
        const char* qg = qhi + gi * qgroup_bytes;
        const float* bg = bound + (size_t)gi * kStreamQ;
        float* tg = tau0 + (size_t)gi * kStreamQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.G * kStreamQ * 4, st));
        hipLaunchKernelGGL((mips_screen_kernel<NKB, 2, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg, nqg,
                           gi * kStreamQ, wgmax, (u64*)nullptr, (int*)nullptr, (int*)nullptr);
        hipLaunchKernelGGL(kth_of_maxima_kernel, dim3(nqg), dim3(256), 0, st, (const unsigned*)wgmax, p.G, k, tg, kStreamQ);
        hipLaunchKernelGGL((mips_screenk_kernel<NKB, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg,
                           (const float*)tg, nqg, cand, cnt, k, sctl);
        hipLaunchKernelGGL((merge_screenk_kernel<BF>), dim3(nqg), dim3(256), merge_lds, st, (const u64*)cand, (const int*)cnt, p.G, k, bg, (const char*)h->hi,
                           (const char*)h->lo, h->nkb, q_dev + (size_t)gi * kStreamQ * h->d, D_dev + (size_t)gi * kStreamQ * k,
                           I_dev + (size_t)gi * kStreamQ * k, id_offset, sctl, kStreamQ, row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// 2 <= k <= 128 with more than 128 queries: groups of 256 on the 32-queries-per-wave kernels
template <bool BF>
int run_screenk32(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, int k, const char* qhi, float* D_dev, long long* I_dev,
                  long long id_offset, hipStream_t st) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    const size_t merge_lds = (size_t)kMergeKLds * 8;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen32_kernel<NKB, 2, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screenk32_kernel<NKB, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)merge_screenk_kernel<BF>, (int)merge_lds);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    float* tau0 = (float*)(ws + p.off_gmax);
    unsigned* wgmax = (unsigned*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    const int ngroups = (nq + kWideQ - 1) / kWideQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const size_t qgroup_bytes = (size_t)kWideQ * h->d * 2;
    MDR_HIP_TRY(hipMemsetAsync(sctl, 0, 256, st));
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kWideQ < kWideQ ? nq - gi * kWideQ : kWideQ;
        const char* qg = qhi + gi * qgroup_bytes;
        const float* bg = bound + (size_t)gi * kWideQ;
        float* tg = tau0 + (size_t)gi * kWideQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.G * kWideQ * 4, st));
        hipLaunchKernelGGL((mips_screen32_kernel<NKB, 2, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg, nqg,
                           gi * kWideQ, wgmax, (u64*)nullptr, (int*)nullptr, (int*)nullptr);
        hipLaunchKernelGGL(kth_of_maxima_kernel, dim3(nqg), dim3(256), 0, st, (const unsigned*)wgmax, p.G, k, tg, kWideQ);
        hipLaunchKernelGGL((mips_screenk32_kernel<NKB, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg,
                           (const float*)tg, nqg, cand, cnt, k, sctl);
        hipLaunchKernelGGL((merge_screenk_kernel<BF>), dim3(nqg), dim3(256), merge_lds, st, (const u64*)cand, (const int*)cnt, p.G, k, bg, (const char*)h->hi,
                           (const char*)h->lo, h->nkb, q_dev + (size_t)gi * kWideQ * h->d, D_dev + (size_t)gi * kWideQ * k,
                           I_dev + (size_t)gi * kWideQ * k, id_offset, sctl, kWideQ, row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

}  // namespace

extern "C" {

int mdr_index_create(int d, int storage, int device, mdr_index** out) {
    MDR_REQUIRE(out != nullptr, "out is NULL");
    MDR_REQUIRE(d > 0 && d % 32 == 0 && d <= 1024, "d=%d unsupported: must be a multiple of 32, <= 1024", d);
    MDR_REQUIRE(storage == MDR_STORE_F32X2H || storage == MDR_STORE_BF16, "unknown storage %d", storage);
    int ndev = 0;
    MDR_HIP_TRY(hipGetDeviceCount(&ndev));
    MDR_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return set_error(MDR_E_HIP, "hipSetDevice(%d) failed", device);
    mdr_index* h = new (std::nothrow) mdr_index();
    MDR_REQUIRE(h != nullptr, "out of host memory");
    h->d = d;
    h->nkb = d / 32;
    h->storage = storage;
    h->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    if (hipMalloc((void**)&h->flags, 256) != hipSuccess || hipMemset(h->flags, 0, 256) != hipSuccess) {
        delete h;
        return set_error(MDR_E_HIP, "hipMalloc(flags) failed");
    }
    *out = h;
    return MDR_OK;
}

int mdr_index_free(mdr_index* h) {
    if (!h) return MDR_OK;
    DeviceGuard g(h->device);
    if (h->hi) (void)hipFree(h->hi);
    if (h->lo) (void)hipFree(h->lo);
    if (h->i8) (void)hipFree(h->i8);
    if (h->centre) (void)hipFree(h->centre);
    if (h->flags) (void)hipFree(h->flags);
    if (h->stage) (void)hipFree(h->stage);
    delete h;
    return MDR_OK;
}

int mdr_index_reserve(mdr_index* h, int64_t n_rows) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(n_rows >= 0 && n_rows < 0xFFFFFFE0ll, "n_rows out of range");
    DeviceGuard g(h->device);
    return grow(h, n_rows, nullptr);
}

int mdr_index_add(mdr_index* h, const void* rows, int64_t n, int src_dtype, int rows_on_device, void* stream) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(n >= 0, "n < 0");
    MDR_REQUIRE(n == 0 || rows != nullptr, "rows is NULL");
    MDR_REQUIRE(src_dtype >= MDR_DT_F32 && src_dtype <= MDR_DT_F16, "unknown src_dtype %d", src_dtype);
    MDR_REQUIRE(h->ntotal + n < 0xFFFFFFE0ll, "index would exceed 2^32-32 rows per shard");
    if (n == 0) return MDR_OK;
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = grow(h, h->ntotal + n, st);
    if (rc) return rc;
    int before[16] = {0};  // [0..3] range / query / norm flags, [8..9] the int8 tier's row statistics: all restored when the rows are rejected
    const bool centre_was_set = h->centre_set;
    MDR_HIP_TRY(hipMemcpyAsync(before, h->flags, sizeof(before), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    const size_t row_src = (size_t)h->d * elem_size(src_dtype);
    if (rows_on_device) {
        rc = add_any(h, rows, src_dtype, n, h->ntotal, st);
        if (rc) return rc;
    } else {
        // chunked H2D through a device staging buffer (<= 256 MiB), converting straight into the shard
        const long long chunk_rows = (long long)((256ull << 20) / row_src);
        size_t need = (size_t)(n < chunk_rows ? n : chunk_rows) * row_src;
        if (need > h->stage_bytes) {
            if (h->stage) MDR_HIP_TRY(hipFree(h->stage));
            h->stage = nullptr;
            h->stage_bytes = 0;
            MDR_HIP_TRY(hipMalloc(&h->stage, need));
            h->stage_bytes = need;
        }
        for (long long r0 = 0; r0 < n; r0 += chunk_rows) {
            long long nr = n - r0 < chunk_rows ? n - r0 : chunk_rows;
            MDR_HIP_TRY(hipMemcpyAsync(h->stage, (const char*)rows + (size_t)r0 * row_src, (size_t)nr * row_src, hipMemcpyHostToDevice, st));
            rc = add_any(h, h->stage, src_dtype, nr, h->ntotal + r0, st);
            if (rc) return rc;
            MDR_HIP_TRY(hipStreamSynchronize(st));  // staging buffer is reused; host buffer must be consumed before return
        }
    }
    int flag = 0;
    MDR_HIP_TRY(hipMemcpyAsync(&flag, h->flags, sizeof(int), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    if (flag) {
        // roll back: the rows stay invisible (ntotal unchanged), the norm bound returns to its previous value
        MDR_HIP_TRY(hipMemcpyAsync(h->flags, before, sizeof(before), hipMemcpyHostToDevice, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        h->centre_set = centre_was_set;  // a centre taken from rejected rows is forgotten: the next accepted add() defines it
        return set_error(MDR_E_RANGE, "add(): a value is non-finite; rows were not added");
    }
    h->ntotal += n;
    return MDR_OK;
}

int64_t mdr_index_ntotal(const mdr_index* h) { return h ? h->ntotal : 0; }
int mdr_index_dim(const mdr_index* h) { return h ? h->d : 0; }
int64_t mdr_index_stream_bytes(const mdr_index* h) { return h ? (int64_t)((h->ntotal + 15) / 16 * 16) * (int64_t)h->d * (is_bf16(h) ? 2 : 4) : 0; }

int mdr_index_set_variant(mdr_index* h, int variant) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(variant >= 0 && variant <= 4, "variant must be 0 (auto), 1 (generic), 2 (exact stream), 3 (screen + refine) or 4 (screen + refine without the int8 tier)");
    h->variant = variant;
    return MDR_OK;
}

const char* mdr_index_last_kernel(const mdr_index* h) { return h ? h->last_kernel : "none"; }

int mdr_index_queries_per_pass(const mdr_index* h, int nq, int k) {
    if (!h || k < 1 || k > kKMax || nq < 1) return 0;
    const int path = make_plan(h, nq, k).path;
    if (path == PATH_GENERIC) return kGenericQ;
    return path == PATH_SCREEN && wide_pass(nq) ? kWideQ : kStreamQ;
}

size_t mdr_index_search_workspace_bytes(const mdr_index* h, int nq, int k) {
    if (!h || nq < 0 || k < 1 || k > kKMax) return 0;
    return make_plan(h, nq, k).total;
}

int mdr_index_search(mdr_index* h, const float* q_dev, int nq, int k, float* D_dev, int64_t* I_dev, int64_t id_offset, void* workspace_dev,
                     size_t workspace_bytes, void* stream) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(nq >= 0, "nq < 0");
    MDR_REQUIRE(k >= 1 && k <= kKMax, "k=%d out of range [1, %d]", k, kKMax);
    if (nq == 0) return MDR_OK;
    MDR_REQUIRE(q_dev && D_dev && I_dev, "NULL query/result pointer");
    if (h->variant == PATH_STREAM && !stream_kernel_supports(h, k))
        return set_error(MDR_E_INVALID, "stream kernel forced but unsupported for d=%d k=%d storage=%d (needs F32X2H, d=768, k<=128)", h->d, k, h->storage);
    if (h->variant == PATH_SCREEN && !screen_kernel_supports(h, k))
        return set_error(MDR_E_INVALID, "screen kernel forced but unsupported for d=%d k=%d (needs d=768, k<=128)", h->d, k);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    long long* I_ll = (long long*)I_dev;
    if (h->ntotal == 0) {
        long long n = (long long)nq * k;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, D_dev, I_ll, n);
        MDR_HIP_TRY(hipGetLastError());
        h->last_kernel = "fill_empty_kernel";
        return MDR_OK;
    }
    SearchPlan p = make_plan(h, nq, k);
    if (!workspace_dev || workspace_bytes < p.total)
        return set_error(MDR_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p.total, workspace_bytes);
    char* ws = (char*)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    const bool bf = is_bf16(h);
    int rc;
    if (p.path == PATH_GENERIC) {
        rc = bf ? run_generic<true>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, nullptr, st)
                : run_generic<false>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, nullptr, st);
        h->last_kernel = "mips_generic_kernel";
        return rc;
    }

    constexpr int NKB = 24;
    const size_t rb_bytes = (size_t)NKB * 2 * kFragBytes;  // one exact-kernel stage == one screen-kernel stage == 48 KiB
    const int n_rb = (int)((h->ntotal + 15) / 16);
    const int ngroups = (nq + kStreamQ - 1) / kStreamQ;
    const size_t qgroup_bytes = (size_t)kStreamQ * h->d * 2;
    u64* best = (u64*)(ws + p.off_best);
    char* qhi = ws + p.off_qhi;
    char* qlo = ws + p.off_qlo;
    float* qscale = (float*)(ws + p.off_qscale);
    {
        // |q.x - qh.xh| <= c |q| max|x|: fp16 rounding of both operands (2^-10) or bf16 rounding of q only (2^-9; the
        // stored rows ARE the bf16 values), plus fp32 accumulation slack
        const float c = bf ? 2.2e-3f : 1.2e-3f;
        const int nq_pad = (nq + kWideQ - 1) / kWideQ * kWideQ;
        float* bound = (float*)(ws + p.off_bound);
        MDR_HIP_TRY(hipMemsetAsync(h->flags + 1, 0, sizeof(int), st));  // "a query of THIS call was non-finite" (telemetry)
        if (bf)
            hipLaunchKernelGGL((prep_queries_kernel<true>), dim3((nq_pad + 3) / 4), dim3(256), 0, st, q_dev, nq, nq_pad, h->d, h->flags, c, qhi, qlo, bound, qscale, 1.0f);
        else
            hipLaunchKernelGGL((prep_queries_kernel<false>), dim3((nq_pad + 3) / 4), dim3(256), 0, st, q_dev, nq, nq_pad, h->d, h->flags, c, qhi, qlo, bound, qscale,
                               row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }

    if (k == 1) {
        MDR_HIP_TRY(hipMemsetAsync(best, 0, (size_t)nq * 8, st));
        const int* run_if = nullptr;
        if (p.path == PATH_SCREEN) {
            if (wide_pass(nq) && p.i8) {  // int8 tier, 256 queries per pass; the fp16 wide screen only behind an overflow
                rc = run_screen8w(h, p, ws, q_dev, nq, best, st);
                if (!rc) rc = run_screen32<false>(h, p, ws, q_dev, nq, qhi, best, st, (const int*)(ws + p.off_ctl8));
                h->last_kernel = "mips_screen8w_kernel<12,1>";
            } else if (wide_pass(nq)) {  // more than 128 queries: 256 per corpus pass on the 32-queries-per-wave kernel
                rc = bf ? run_screen32<true>(h, p, ws, q_dev, nq, qhi, best, st) : run_screen32<false>(h, p, ws, q_dev, nq, qhi, best, st);
                h->last_kernel = bf ? "mips_screen32_kernel<24,1,bf16>" : "mips_screen32_kernel<24,1>";
            } else if (p.i8) {  // int8 tier first; the fp16 screen only if one of its lists overflowed, the exact pass only if that one's did
                rc = run_screen8(h, p, ws, q_dev, nq, best, st);
                if (!rc) rc = run_screen<false>(h, p, ws, q_dev, nq, qhi, best, st, (const int*)(ws + p.off_ctl8));
                h->last_kernel = "mips_screen8_kernel<12,1>";
            } else {
                rc = bf ? run_screen<true>(h, p, ws, q_dev, nq, qhi, best, st) : run_screen<false>(h, p, ws, q_dev, nq, qhi, best, st);
                h->last_kernel = bf ? "mips_screen_kernel<24,1,bf16>" : "mips_screen_kernel<24,1>";
            }
            if (rc) return rc;
            run_if = (const int*)(ws + p.off_sctl);  // exact pass below: only if a candidate list overflowed
        } else {
            h->last_kernel = "mips_stream_kernel<24,0>";
        }
        if (!bf) {
            rc = ensure_dynamic_lds((const void*)mips_stream_kernel<NKB, 0>, (int)(3 * rb_bytes));
            if (rc) return rc;
            const int Gx = (int)(n_rb < h->num_cus ? n_rb : h->num_cus);
            for (int gi = 0; gi < ngroups; ++gi) {
                const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
                hipLaunchKernelGGL((mips_stream_kernel<NKB, 0>), dim3(Gx), dim3(512), 3 * rb_bytes, st, (const char*)h->hi, (const char*)h->lo,
                                   (long long)h->ntotal, n_rb, (const char*)(qhi + gi * qgroup_bytes), (const char*)(qlo + gi * qgroup_bytes), nqg,
                                   best + (size_t)gi * kStreamQ, (u64*)nullptr, (int*)nullptr, (u64*)nullptr, 1, run_if,
                                   (const float*)(qscale + (size_t)gi * kStreamQ));
                MDR_HIP_TRY(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(finalize_top1_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, (const u64*)best, nq, D_dev, I_ll, (long long)id_offset);
        MDR_HIP_TRY(hipGetLastError());
        if (bf)  // bf16 storage has no exact MFMA pass: the overflow fallback is the generic kernel, overwriting D/I
            return run_generic<true>(h, p, ws, q_dev, nq, 1, D_dev, I_ll, id_offset, run_if, st);
        return MDR_OK;
    }

    // 2 <= k <= 128
    const int* run_if = nullptr;
    if (p.path == PATH_SCREEN) {
        if (wide_pass(nq)) {
            rc = bf ? run_screenk32<true>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st)
                    : run_screenk32<false>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st);
            h->last_kernel = bf ? "mips_screenk32_kernel<24,bf16>" : "mips_screenk32_kernel<24>";
        } else {
            rc = bf ? run_screenk<true>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st)
                    : run_screenk<false>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st);
            h->last_kernel = bf ? "mips_screenk_kernel<24,bf16>" : "mips_screenk_kernel<24>";
        }
        if (rc) return rc;
        run_if = (const int*)(ws + p.off_sctl);  // exact pass below: only if a list or the band overflowed
        if (bf) return run_generic<true>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, run_if, st);
    } else {
        h->last_kernel = "mips_stream_kernel<24,1>";
    }
    // F32X2H: exact stream kernel with candidate lists (unconditional on the stream path)
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    u64* kth = (u64*)(ws + p.off_kth);
    const size_t lds_bytes = 3 * rb_bytes + kStreamQ * sizeof(int);
    rc = ensure_dynamic_lds((const void*)mips_stream_kernel<NKB, 1>, (int)lds_bytes);
    if (rc) return rc;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.Gx * kStreamQ * 4, st));
        MDR_HIP_TRY(hipMemsetAsync(kth, 0, (size_t)p.Gx * kStreamQ * 8, st));
        hipLaunchKernelGGL((mips_stream_kernel<NKB, 1>), dim3(p.Gx), dim3(512), lds_bytes, st, (const char*)h->hi, (const char*)h->lo, (long long)h->ntotal,
                           n_rb, (const char*)(qhi + gi * qgroup_bytes), (const char*)(qlo + gi * qgroup_bytes), nqg, (u64*)nullptr, cand, cnt, kth, k,
                           run_if, (const float*)(qscale + (size_t)gi * kStreamQ));
        MDR_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(merge_lists_kernel, dim3(nqg), dim3(256), 0, st, (const u64*)cand, (const int*)cnt, (const u64*)kth, p.Gx, kStreamQ, kStreamCap, k,
                           D_dev + (size_t)gi * kStreamQ * k, I_ll + (size_t)gi * kStreamQ * k, (long long)id_offset, run_if,
                           (const float*)(qscale + (size_t)gi * kStreamQ));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

#if MDR_I8_ABL == 9  // measurement builds only (include/mdr_hip_measure.h)
int mdr_test_i8_stamps(unsigned long long* out8_host, int reset) {
    MDR_REQUIRE(out8_host, "NULL pointer");
    MDR_HIP_TRY(hipDeviceSynchronize());
    MDR_HIP_TRY(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_i8_stamp), 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        MDR_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamp), z, sizeof(z)));
    }
    return MDR_OK;
}
#endif

int mdr_index_search_telemetry(const mdr_index* h, int nq, int k, const void* workspace_dev, int64_t* out4_host, void* stream) {
    MDR_REQUIRE(h && workspace_dev && out4_host, "NULL argument");
    MDR_REQUIRE(nq >= 1 && k >= 1 && k <= kKMax, "bad shape");
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    const SearchPlan p = make_plan(h, nq, k);
    const char* ws = (const char*)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    out4_host[0] = out4_host[1] = out4_host[2] = 0;
    out4_host[3] = p.path;
    int flags[4] = {0, 0, 0, 0};
    MDR_HIP_TRY(hipMemcpyAsync(flags, h->flags, sizeof(flags), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    out4_host[2] = flags[1];
    if (p.path != PATH_SCREEN) return MDR_OK;
    int overflow = 0;
    MDR_HIP_TRY(hipMemcpyAsync(&overflow, ws + p.off_sctl, sizeof(int), hipMemcpyDeviceToHost, st));
    long long total = 0;
    if (k == 1) {  // per-wave list lengths of the last query group
        const size_t n_cnt = (size_t)(p.i8 ? (wide_pass(nq) ? p.G8w : p.G8) : p.G) * 8;
        int* cnt = new (std::nothrow) int[n_cnt];
        MDR_REQUIRE(cnt != nullptr, "out of host memory");
        hipError_t e = hipMemcpyAsync(cnt, ws + p.off_sctl + 256, n_cnt * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        for (size_t i = 0; i < n_cnt; ++i) total += cnt[i];
        delete[] cnt;
        if (e != hipSuccess) return set_error(MDR_E_HIP, "telemetry copy failed: %s", hipGetErrorString(e));
    } else {  // merge_screenk_kernel adds every query's union size to sctl[1]
        int t = 0;
        MDR_HIP_TRY(hipMemcpyAsync(&t, ws + p.off_sctl + sizeof(int), sizeof(int), hipMemcpyDeviceToHost, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        total = t;
    }
    out4_host[0] = overflow;
    out4_host[1] = total;
    if (p.i8) {  // bit 9: the int8 tier ran in front; bit 8: one of its lists overflowed (the fp16 screen ran behind it)
        int o8 = 0;
        MDR_HIP_TRY(hipMemcpyAsync(&o8, ws + p.off_ctl8, sizeof(int), hipMemcpyDeviceToHost, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        int kept = 0;
        MDR_HIP_TRY(hipMemcpyAsync(&kept, ws + p.off_ctl8 + sizeof(int), sizeof(int), hipMemcpyDeviceToHost, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        out4_host[3] |= 512 | (o8 ? 256 : 0) | ((int64_t)kept << 16);  // bits 16..: candidates of the int8 tier that were really re-scored
    }
    return MDR_OK;
}

int mdr_topk_merge(const float* D_parts_dev, const int64_t* I_parts_dev, int nparts, int nq, int k, float* D_dev, int64_t* I_dev, void* stream) {
    MDR_REQUIRE(nparts >= 1 && nq >= 0 && k >= 1 && k <= kKMax, "bad merge shape nparts=%d nq=%d k=%d", nparts, nq, k);
    if (nq == 0) return MDR_OK;
    MDR_REQUIRE(D_parts_dev && I_parts_dev && D_dev && I_dev, "NULL pointer");
    hipLaunchKernelGGL(merge_parts_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, D_parts_dev, (const long long*)I_parts_dev, nparts, nq, k, D_dev,
                       (long long*)I_dev);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

}  // extern "C"

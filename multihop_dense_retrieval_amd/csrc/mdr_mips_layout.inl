// csrc/mdr_mips_layout.inl -- types, order-preserving keys, fragment-tiled addressing, add()-side kernels (fp16 hi/lo conversion with the index exponent,
// row norms, absmax, rescale) and the fp16 query preparation. Included by mdr_mips.hip inside namespace mdr::{anonymous}; not a translation unit of its own.
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
__device__ __forceinline__ void prep_queries_body(int bx, const float* __restrict__ q, int nq, int nq_pad, int d, int* __restrict__ flags, float c,
                                                  char* __restrict__ qhi, char* __restrict__ qlo, float* __restrict__ bound,
                                                  float* __restrict__ qscale, float xs /* 2^E of the stored rows: folded into qscale */) {
    const int lane = threadIdx.x & 63;
    const int i = bx * 4 + (threadIdx.x >> 6);
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

template <bool BF>
__global__ void __launch_bounds__(256) prep_queries_kernel(const float* __restrict__ q, int nq, int nq_pad, int d, int* __restrict__ flags, float c,
                                                           char* __restrict__ qhi, char* __restrict__ qlo, float* __restrict__ bound,
                                                           float* __restrict__ qscale, float xs) {
    prep_queries_body<BF>((int)blockIdx.x, q, nq, nq_pad, d, flags, c, qhi, qlo, bound, qscale, xs);
}

__device__ inline int block_sum_256(int v, int* red) {
    // red: LDS int[4]
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

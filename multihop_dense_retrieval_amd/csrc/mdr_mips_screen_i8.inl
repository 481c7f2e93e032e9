// csrc/mdr_mips_screen_i8.inl -- the int8 screening tier of the k = 1 search: centred / column-weighted plane (add() side), query quantisation with rigorous bounds,
// the 16- and 32-queries-per-wave kernels, star-row and candidate re-scoring. Included by mdr_mips.hip inside namespace mdr::{anonymous}.
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
// Query split (round 6; `CB` instantiations, chosen per index when its centre is large against the rows' spread around it -- real embedding matrices): a query of
// the same kind as the rows carries the common component too, q = lambda_q c + dq with lambda_q = (q.c) / (c.c), and quantised whole it spends its 8 bits on
// lambda_q c (which says nothing about WHICH row wins) instead of on dq. Since
//     q.(x_r - c) = dq.(x_r - c) + lambda_q b_r,     b_r = c.(x_r - c)  (one fp32 per row, kept beside the row scale in the super-block's tail: no extra bytes),
// the plane is scored against dq (t, alpha, beta now come from dq: several times smaller) and the per-row term enters the bounds exactly:
//     U_r = s_r (t A + alpha_q) + beta_q + lambda_q b_r,   L_r = U_r - 2 (alpha_q s_r + beta_q).
// ANY lambda is correct; fp32 rounding of b_r (768 terms) is covered by |lambda_q| 1e-4 max_r sum_i |c_i (x_ri - c_i)| (i8stats[2]) added to beta_q.
// Measured at 5 M rows on this repo's encoder outputs (|c| = 26.5, centred rows 9.8): see DESIGN.md section 4.
// Layout: a super-block (32 rows) = 2 x NKB8 fragment blocks of 1 KiB (16 rows x 64 int8; lane (lr, g) of the MFMA owns the
// 16 bytes k = 64 kb + 16 g .. of row lr at (16 g + lr) * 16) followed by 256 bytes holding the 32 row scales and the 32 row terms b_r: 24.25 KiB at
// d = 768 against the 48 KiB of the fp16 hi plane.
#ifndef MDR_I8_ABL
#define MDR_I8_ABL 0  // measurement builds (wrong results): 1 no scale-tail DMA, 2 no epilogue, 3 no MFMAs, 4 no fragment reads
#endif
constexpr int kI8RefinePerQuery = 8192;  // emitted candidates per query of a pass beyond which the int8 tier hands over to the fp16 screen (see mips_refine8_kernel)
constexpr int kI8Tail = 256;  // bytes behind a super-block's fragments: 32 fp32 row scales, then 32 fp32 row terms b_r = c.(x_r - c): one 4-byte-per-lane DMA piece
constexpr int kI8TailB = 128;  // offset of the row terms inside the tail
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
// *cb_flag = 1 when the centre is large against the spread around it (|c|^2 >= kCbRatio2 x the mean squared distance of a row from c): the query split pays.
constexpr float kCbRatio2 = 0.02f;
__global__ void __launch_bounds__(1024) centre_finish_kernel(const float* __restrict__ sums, int d, float inv_n, float* __restrict__ cw, int* __restrict__ cb_flag) {
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
    const float var_sum = block_reduce(var, false);
    const float mu2_sum = block_reduce(mu * mu, false);
    if (i == 0) *cb_flag = (mu2_sum >= kCbRatio2 * var_sum && mu2_sum > 0.f) ? 1 : 0;
    const float ref = sqrtf(var_sum / (float)d);
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

// one wave per row: lanes 0 .. d/16-1 quantise 16 consecutive columns each of x - centre. stats[0] = max s_r, stats[1] = max s_r (L1(x8_r)/2 + d/4),
// stats[2] = max_r sum_i |c_i (x_ri - c_i)| (what the fp32 rounding of the row term b_r = c.(x_r - c) scales with)
// (non-negative floats, kept as their bit patterns: they order like ints)
template <typename T>
__global__ void __launch_bounds__(256) convert_to_i8_kernel(const T* __restrict__ src, long long n, int d, long long row0, char* __restrict__ dst,
                                                            int* __restrict__ stats, const float* __restrict__ centre) {
    const int lane = threadIdx.x & 63;
    const int nkb8 = d >> 6;
    const bool on = lane < (d >> 4);
    float cj[16], iw[16];  // this lane's 16 columns of the centre and of 1 / w: loaded once, the wave then walks its rows
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        cj[j] = on ? centre[lane * 16 + j] : 0.f;
        iw[j] = on ? centre[d + lane * 16 + j] : 0.f;
    }
    float smax = 0.f, cmax = 0.f, bamax = 0.f;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (long long)gridDim.x * 4) {
        float x[16];
        float mx = 0.f, br = 0.f, bra = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float xc = on ? load_as_f32<T>(src + r * (long long)d + lane * 16 + j) - cj[j] : 0.f;
            br = fmaf(cj[j], xc, br);
            bra = fmaf(fabsf(cj[j]), fabsf(xc), bra);
            x[j] = xc * iw[j];  // (x - c) / w
            mx = fmaxf(mx, fabsf(x[j]));
        }
        mx = wave_max_f(mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { br += __shfl_xor(br, o); bra += __shfl_xor(bra, o); }
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
            *(float*)(sb + (size_t)2 * nkb8 * kFragBytes + kI8TailB + (row & 31) * 4) = br;
        }
        bamax = fmaxf(bamax, bra);
        smax = fmaxf(smax, sc);
        cmax = fmaxf(cmax, sc * (0.5f * (float)l1 + 0.25f * (float)d));
    }
    if (lane == 0) {  // one pair of atomics per wave instead of per row
        if (__float_as_int(smax) > stats[0]) atomicMax(stats + 0, __float_as_int(smax));
        if (__float_as_int(cmax) > stats[1]) atomicMax(stats + 1, __float_as_int(cmax));
        if (__float_as_int(bamax) > stats[2]) atomicMax(stats + 2, __float_as_int(bamax));
    }
}

// one wave per query row (rows >= nq: zero padding). q8: fragment-tiled like the corpus blocks (16 queries per block, NKB8 KiB
// each); qab[i] = (t, alpha, beta, 0) with the 1e-3 inflation described above.
// qab[i][3] = q.c + slack (c = the plane's centre): what is subtracted from an EXACT score of a row to get a valid lower bound of its
// centred score q.(x - c). slack = 1e-4 sum|q_i c_i| + 2e-6 |q.c| covers the fp32 summation of q.c (768 terms) and the fp32 rounding of
// x - c in convert_to_i8_kernel; the 1e-3 inflation of alpha / beta covers the rest as before.
// cb != 0 (the index uses the query split): the plane is scored against dq = q - lambda c, lambda = (q.c) / (c.c) -> qlam[i]; cb == 0: lambda = 0, dq = q.
__device__ __forceinline__ void prep_queries_i8_body(int bx, const float* __restrict__ q, int nq, int nq_pad, int d, const int* __restrict__ stats,
                                                     char* __restrict__ q8, f32x4* __restrict__ qab, const float* __restrict__ centre, int cb,
                                                     float* __restrict__ qlam) {
    const int lane = threadIdx.x & 63;
    const int i = bx * 4 + (threadIdx.x >> 6);
    if (i >= nq_pad) return;
    const int nkb8 = d >> 6;
    const bool on = lane < (d >> 4) && i < nq;
    float x[16], cv[16];
    float mx = 0.f, qc = 0.f, qca = 0.f, cc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x[j] = on ? q[(size_t)i * d + lane * 16 + j] : 0.f;
        cv[j] = on ? centre[lane * 16 + j] : 0.f;
        qc = fmaf(x[j], cv[j], qc);
        qca = fmaf(fabsf(x[j]), fabsf(cv[j]), qca);
        cc = fmaf(cv[j], cv[j], cc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qc += __shfl_xor(qc, o); qca += __shfl_xor(qca, o); cc += __shfl_xor(cc, o); }
    float lam = cb && cc > 0.f ? qc / cc : 0.f;
    if (!(fabsf(lam) <= 3.0e38f)) lam = 0.f;  // (a non-finite query: the bounds below become infinite anyway)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x[j] = fmaf(-lam, cv[j], x[j]) * (on ? centre[2 * d + lane * 16 + j] : 0.f);  // (q_i - lambda c_i) w_i (w a power of two: exact)
        mx = fmaxf(mx, fabsf(x[j]));
    }
    mx = wave_max_f(mx);
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
        const float s2 = __int_as_float(stats[1]), s3 = __int_as_float(stats[2]);
        f32x4 o = {t, 0.5f * t * (float)l1 * 1.001f, t * s2 * 1.001f + fabsf(lam) * 1e-4f * s3, qc + (1e-4f * qca + 2e-6f * fabsf(qc))};
        if (!fin) o = (f32x4){0.f, INFINITY, INFINITY, 0.f};
        qab[i] = o;
        qlam[i] = fin ? lam : 0.f;
    }
}

// ONE launch for both query preparations of a k = 1 search over an fp32-accurate index with an int8 plane (round 5: one kernel boundary less per call):
// blocks [0, nb16) = prep_queries_body<false> (fp16 hi / lo fragments, bounds, scales), blocks [nb16, nb16 + nb8) = prep_queries_i8_body.
__global__ void __launch_bounds__(256) prep_queries_both_kernel(int nb16, const float* __restrict__ q, int nq, int nq_pad16, int nq_pad8, int d, int* __restrict__ flags,
                                                                float c, char* __restrict__ qhi, char* __restrict__ qlo, float* __restrict__ bound,
                                                                float* __restrict__ qscale, float xs, char* __restrict__ q8, f32x4* __restrict__ qab,
                                                                const float* __restrict__ centre, int cb, float* __restrict__ qlam) {
    if ((int)blockIdx.x < nb16) prep_queries_body<false>((int)blockIdx.x, q, nq, nq_pad16, d, flags, c, qhi, qlo, bound, qscale, xs);
    else prep_queries_i8_body((int)blockIdx.x - nb16, q, nq, nq_pad8, d, (const int*)(flags + 8), q8, qab, centre, cb, qlam);
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
template <int NKB8, int MODE, int NS, bool CB = false>  // NS: LDS slots of one super-block (NS - 1 stages in flight); CB: the query split (header of this file)
__global__ void __launch_bounds__(512, 2)
mips_screen8_kernel(const char* __restrict__ X8, long long n_rows, int n_sb, const char* __restrict__ Q8, const f32x4* __restrict__ qab, int nq, int q_base,
                    unsigned* __restrict__ gmax /* [nq] ordered(max L) */, u64* __restrict__ cand /* [waves][kWaveCandCap] */,
                    int* __restrict__ cand_cnt /* [waves] */, int* __restrict__ overflow, u64* __restrict__ gstar /* [nq] (ordered max L, its row) */,
                    const u64* __restrict__ best /* MODE 1: exact keys of the sample pass's star rows (a tighter first `known`) */,
                    const float* __restrict__ qlam /* CB: lambda_q */) {
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
    float ql = CB && q_valid ? qlam[qlocal] : 0.f;
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
    asm volatile("" : "+v"(qt), "+v"(qa), "+v"(qb), "+v"(known), "+v"(ql));
    const f32x2 qt2 = {qt, qt}, qa2 = {qa, qa}, qb2 = {qb, qb}, ql2 = {ql, ql};
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
        f32x4 br0 = {0.f, 0.f, 0.f, 0.f}, br1 = br0;  // CB: this lane's 8 row terms b_r
        if (CB) {
            br0 = *(const f32x4*)(slot + 2 * NKB8 * kFragBytes + kI8TailB + sub_row * 4);
            br1 = *(const f32x4*)(slot + 2 * NKB8 * kFragBytes + kI8TailB + (16 + sub_row) * 4);
        }
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
            f32x2 ad0 = qb2, ad1 = qb2;  // beta_q (+ lambda_q b_r)
            if (CB) {
                ad0 = __builtin_elementwise_fma((f32x2){br0[2 * pr], br0[2 * pr + 1]}, ql2, qb2);
                ad1 = __builtin_elementwise_fma((f32x2){br1[2 * pr], br1[2 * pr + 1]}, ql2, qb2);
            }
            u2[pr] = __builtin_elementwise_fma(__builtin_elementwise_fma(f0, qt2, qa2), s0, ad0);
            u2[2 + pr] = __builtin_elementwise_fma(__builtin_elementwise_fma(f1, qt2, qa2), s1, ad1);
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
#ifndef MDR_I8W_PIN
#define MDR_I8W_PIN 1  // 1 (product since round 6): pin every `bounds` slice of the wide kernel behind its MFMA; 0 (measurement): let hipcc sink them into `decide`
#endif
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

template <int NKB8, int MODE, int NS, bool CB = false>
__global__ void __launch_bounds__(512, 2)
mips_screen8w_kernel(const char* __restrict__ X8, long long n_rows, int n_sb, const char* __restrict__ Q8, const f32x4* __restrict__ qab, int nq, int q_base,
                     unsigned* __restrict__ gmax, u64* __restrict__ cand, int* __restrict__ cand_cnt, int* __restrict__ overflow, u64* __restrict__ gstar,
                     const u64* __restrict__ best, const float* __restrict__ qlam /* CB: lambda_q */) {
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
    float ql = CB && q_valid ? qlam[qlocal] : 0.f;
    float known = -FLT_MAX;
    if (MODE == 1 && q_valid) {
        unsigned g = gmax[qlocal];
        if (g) known = unord32(g);
        const u64 kb = best[q_base + qlocal];
        if (kb) known = fmaxf(known, key_score(kb) - ab[3]);  // exact score -> centred lower bound (see mips_screen8_kernel)
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) asm volatile("" : "+v"(qf[sl]));
    asm volatile("" : "+v"(qt), "+v"(qa), "+v"(qb), "+v"(known), "+v"(ql));
    const f32x2 qt2 = {qt, qt}, qa2 = {qa, qa}, qb2 = {qb, qb}, ql2 = {ql, ql};
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
    // CB: the addends beta_q + lambda_q b_r of the 16 rows of the super-block whose epilogue is PENDING, two per register pair. One set serves both accumulator
    // sets (a second one spilled: 266 VGPRs). Its row terms b_r are read from the super-block's own LDS slot behind MFMA 9 of that super-block's chain -- the
    // previous super-block's epilogue (MFMAs 0-8) has used the old values by then, and the slot is still this stage's -- by inline-asm reads the chain's own
    // counted waits cover (LDS returns in order: behind MFMA 13's wait they have landed), and turned into addends behind MFMAs 13-20, one packed FMA each, where
    // the chain has no other VALU work. The first version did the FMA inside `bounds` (MFMAs 0-7, already the busiest steps): +0.22 ms per 256-query pass.
    static_assert(!CB || (MDR_I8W_PF <= 4 && 2 * NKB8 >= 22), "the b_r reads issued behind MFMA 9 are covered by the chain's wait before MFMA 13 only with <= 4 reads in flight");
    f32x2 adv[CB ? 8 : 1];
    f32x4 brs[CB ? 4 : 1];
#pragma unroll
    for (int j = 0; j < (CB ? 8 : 1); ++j) adv[j] = qb2;
#pragma unroll
    for (int j = 0; j < (CB ? 4 : 1); ++j) brs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x2 u2[8];
    float mu = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < 8; ++j) u2[j] = (f32x2){0.f, 0.f};
    auto bounds = [&](const Pending& R, auto jc) __attribute__((always_inline)) {  // accumulators 2 j, 2 j + 1 of super-block R
        constexpr int j = decltype(jc)::value;
        const f32x2 f = {(float)R.acc[2 * j], (float)R.acc[2 * j + 1]};
        const f32x2 sc = {R.sr[j >> 1][2 * (j & 1)], R.sr[j >> 1][2 * (j & 1) + 1]};
        f32x2 ad = qb2;  // beta_q (+ lambda_q b_r)
        if constexpr (CB) ad = adv[j];
        u2[j] = __builtin_elementwise_fma(__builtin_elementwise_fma(f, qt2, qa2), sc, ad);
        mu = j == 0 ? fmaxf(u2[0][0], u2[0][1]) : fmaxf(mu, fmaxf(u2[j][0], u2[j][1]));
#if MDR_I8W_PIN
        // Round 6: without this pin hipcc SINKS the eight `bounds` slices into `decide` (their only consumer, under `if (R.have)`): the ISA showed MFMAs 0-8 back to back and
        // then one block of 16 conversions + 16 packed FMAs + 25 max operations between MFMA 8 and MFMA 9 -- the epilogue was inside the chain but not IN THE SHADOW of its
        // MFMAs. An opaque use keeps slice j behind MFMA j (the sched_barrier behind every fill() only orders what is already there). Measured, same box, alternating
        // (profiles/r06_i8w_pinned_bounds_ab.txt): main pass 982-987 -> 974 us, search stage 1.094-1.100 -> 1.076-1.084 ms, 244 instead of 234 VGPRs. Two waves per SIMD
        // were already hiding most of the block; the gain is what they did not.
        asm volatile("" : "+v"(u2[j]), "+v"(mu));
#endif
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
                        else if constexpr (CB && sl == 9) {
                            const unsigned ta = (unsigned)(uintptr_t)(slot + 2 * NKB8 * kFragBytes + kI8TailB + 16 * lh);
                            asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(brs[0]) : "v"(ta));
                            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(brs[1]) : "v"(ta));
                            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(brs[2]) : "v"(ta));
                            asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(brs[3]) : "v"(ta));
                        } else if constexpr (CB && sl >= 13 && sl < 21) {
                            constexpr int j = sl - 13;
                            asm volatile("" : "+v"(brs[j >> 1]));  // (the value the asm read above delivered: not to be folded with the old one)
                            adv[j] = __builtin_elementwise_fma((f32x2){brs[j >> 1][2 * (j & 1)], brs[j >> 1][2 * (j & 1) + 1]}, ql2, qb2);
                        }
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
    // Round 5: FILTER first, all 256 threads one candidate each, survivors compacted into LDS; THEN the 16-lane groups re-score survivors only. Before,
    // the groups walked the list 16 candidates at a time and an iteration took an exact re-scoring (a chain of dependent loads, ~2.5 us) whenever ONE of
    // its 16 candidates survived the filters -- ~1 in 8 does, so nearly every iteration did: a 200-entry list cost 13 such rounds (31 us per launch in the
    // pipelined loop's profile) where its ~25 survivors need two.
    __shared__ unsigned short surv[kWaveCandCap];
    __shared__ int n_surv;
    if (threadIdx.x == 0) n_surv = 0;
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 256) {
        const u64 e = list[c];
        const unsigned qi = (unsigned)(e >> 48), u16 = (unsigned)(e >> 32) & 0xFFFFu;
        if (u16 < (gmax[qi] >> 16)) continue;  // U < final max L: cannot be the best row
        if (u16 < thr[qi & 31]) continue;
        surv[atomicAdd(&n_surv, 1)] = (unsigned short)c;  // (n <= kWaveCandCap: cannot overflow)
    }
    __syncthreads();
    const int ns = n_surv;
    int kept = 0;
    for (int s_ = threadIdx.x >> 4; s_ < ns; s_ += 16) {
        const u64 e = list[surv[s_]];
        const unsigned qi = (unsigned)(e >> 48), u16 = (unsigned)(e >> 32) & 0xFFFFu, row = (unsigned)e;
        if (u16 < thr[qi & 31]) continue;  // (raised meanwhile by this block's own re-scorings)
        const float acc = exact_dot16<false>(Xhi, Xlo, nkb, q + (size_t)qi * d, row, sub, xs);
        if (sub == 0) {
            atomicMax(best + qi, make_key(acc, row));
            atomicMax(&thr[qi & 31], ord32(acc - qoff[qi & 31]) >> 16);
            ++kept;
        }
    }
    if (sub == 0 && kept) atomicAdd(ctl8 + 1, kept);
}

// csrc/mdr_encoder_pack_ln.inl -- un-padded packing (lengths, scan, scatter), embedding gather + LayerNorm, the LayerNorm kernel, CLS gather, f32 -> f16.
// Included by mdr_encoder.hip inside namespace mdr::{anonymous}; not a translation unit of its own.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MDR_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MDR_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- packing ---------------------------------------------------------------------------------------
// one wave per row: number of tokens with mask != 0
__global__ void __launch_bounds__(256) enc_lens_kernel(const long long* __restrict__ mask, int B, int L, int* __restrict__ lens) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int n = 0;
    for (int p0 = 0; p0 < L; p0 += 64) {
        int p = p0 + lane;
        bool m = p < L && mask[(size_t)b * L + p] != 0;
        n += __popcll(__ballot(m));
    }
    if (lane == 0) lens[b] = n;
}

// single block: exclusive scan of lens -> cu[0..B], total; and (order != nullptr, B <= 1024) the sequences sorted by length, longest first (ties: lower
// index first) -> order[0..B): the attention kernel for L > 128 walks its (sequence, head) pairs in that order, so that the workgroups that start last are
// the short ones (its timeline showed a quarter of the kernel's span draining 12-us workgroups of long sequences that were dispatched last).
__global__ void __launch_bounds__(1024) enc_scan_kernel(const int* __restrict__ lens, int B, int* __restrict__ cu, int* __restrict__ total, int* __restrict__ order) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    __shared__ unsigned skey[1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        int i = base + tid;
        int v = i < B ? lens[i] : 0;
        int x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int off = carry_s;
        for (int j = 0; j < w; ++j) off += wsum[j];
        if (i < B) cu[i] = off + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (tid == 0) { cu[B] = carry_s; *total = carry_s; }
    if (order == nullptr || B > 1024) return;
    // bitonic sort, descending, of (len << 10 | 1023 - index) over the next power of two (padding keys 0 sink to the end)
    int n2 = 1;
    while (n2 < B) n2 <<= 1;
    skey[tid] = tid < B ? ((unsigned)lens[tid] << 10) | (unsigned)(1023 - tid) : 0u;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = tid ^ j;
            if (tid < n2 && p > tid) {
                const unsigned a = skey[tid], b = skey[p];
                const bool desc = (tid & k) == 0;  // this stretch is sorted descending
                if (desc ? a < b : a > b) { skey[tid] = b; skey[p] = a; }
            }
            __syncthreads();
        }
    if (tid < B) order[tid] = 1023 - (int)(skey[tid] & 1023u);
}

// one wave per row: packed token t = cu[b] + j  ->  source element b*L+p and RoBERTa position id
// (position ids count input_ids != pad_id over the WHOLE row, HF create_position_ids_from_input_ids)
__global__ void __launch_bounds__(256) enc_scatter_kernel(const long long* __restrict__ ids, const long long* __restrict__ mask, int B, int L,
                                                          int pad_id, const int* __restrict__ cu, int* __restrict__ tok_src,
                                                          int* __restrict__ tok_pid) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int j0 = cu[b], c0 = 0;
    for (int p0 = 0; p0 < L; p0 += 64) {
        int p = p0 + lane;
        bool in = p < L;
        bool m = in && mask[(size_t)b * L + p] != 0;
        bool np = in && ids[(size_t)b * L + p] != (long long)pad_id;
        unsigned long long bm = __ballot(m), bn = __ballot(np);
        if (m) {
            int t = j0 + __popcll(bm & lt);
            tok_src[t] = b * L + p;
            tok_pid[t] = np ? (c0 + __popcll(bn & lt) + 1 + pad_id) : pad_id;
        }
        j0 += __popcll(bm);
        c0 += __popcll(bn);
    }
}

// ---- embeddings + LayerNorm: one wave per packed token ------------------------------------------------
constexpr int kMaxPerLane = 16;  // hidden <= 1024

__global__ void __launch_bounds__(256)
embed_ln_kernel(const long long* __restrict__ ids, const int* __restrict__ tok_src, const int* __restrict__ tok_pid, const int* __restrict__ total,
                const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type0, const float* __restrict__ g,
                const float* __restrict__ bta, int H, int vocab, int max_pos, float eps, _Float16* __restrict__ out, float* __restrict__ out32) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= *total) return;
    long long id = ids[tok_src[t]];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int pid = tok_pid[t];
    pid = pid >= max_pos ? max_pos - 1 : pid;
    const float* wr = word + (size_t)id * H;
    const float* pr = pos + (size_t)pid * H;
    const int n = H >> 6;
    float x[kMaxPerLane];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i)
        if (i < n) { int e = lane + 64 * i; x[i] = wr[e] + pr[e] + type0[e]; s += x[i]; }
    const float mu = wave_sum(s) / H;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i)
        if (i < n) { float dlt = x[i] - mu; v += dlt * dlt; }
    const float rstd = rsqrtf(wave_sum(v) / H + eps);
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i)
        if (i < n) {
            const int e = lane + 64 * i;
            const float y = (x[i] - mu) * rstd * g[e] + bta[e];
            out[(size_t)t * H + e] = (_Float16)y;
            if (out32) out32[(size_t)t * H + e] = y;  // fp32 residual stream (mdr_encoder_config.residual_fp32)
        }
}

// fp32 rows (+ residual) -> LayerNorm -> fp16 (out16: the next GEMM's operand) and/or fp32 (out32: the residual stream in
// residual_fp32 mode, or the final embedding); one wave per row, 16-byte loads (H % 256 == 0 fast path).
// `res16` / `res32` (at most one): residual added before normalising, when the producing GEMM left it out. out32 may alias
// res32 (a wave reads its whole row before it writes it).
// IN_T = float: the producing GEMM left its sums in fp32; IN_T = _Float16 (mdr_encoder_config.residual_fp32 = 2): the Linear's output was rounded to
// fp16 as apex O1's F.linear does, half the bytes in and out of the GEMM epilogue.
template <typename IN_T>
__global__ void __launch_bounds__(256)
layernorm_kernel(const IN_T* __restrict__ in, const _Float16* __restrict__ res16, const float* res32, int rows_cap, const int* __restrict__ rows_dev,
                 int H, const float* __restrict__ g, const float* __restrict__ bta, float eps, _Float16* __restrict__ out16, float* out32) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = rows_dev ? min(*rows_dev, rows_cap) : rows_cap;
    if (t >= rows) return;
    const IN_T* r = in + (size_t)t * H;
    if ((H & 255) == 0) {
        const int n4 = H >> 8;  // float4 per lane (<= 4)
        f32x4 x[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n4) {
#if defined(MDR_LN_ABL) && (MDR_LN_ABL == 1 || MDR_LN_ABL == 2)  // measurement builds (results wrong): the Linear's sums are not read -- what a fused GEMM epilogue would save on this side
                x[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#else
                if constexpr (std::is_same<IN_T, float>::value) {
                    x[i] = *(const f32x4*)(r + (lane + 64 * i) * 4);
                } else {
                    const half4 h4 = *(const half4*)(r + (lane + 64 * i) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[i][j] = (float)h4[j];
                }
#endif
                if (res16) {
                    const half4 r4 = *(const half4*)(res16 + (size_t)t * H + (lane + 64 * i) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[i][j] += (float)r4[j];
                }
                if (res32) x[i] += *(const f32x4*)(res32 + (size_t)t * H + (lane + 64 * i) * 4);
                s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
            }
        const float mu = wave_sum(s) / H;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float dlt = x[i][j] - mu; v += dlt * dlt; }
            }
        const float rstd = rsqrtf(wave_sum(v) / H + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n4) {
                const int e = (lane + 64 * i) * 4;
                const f32x4 g4 = *(const f32x4*)(g + e), b4 = *(const f32x4*)(bta + e);
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (x[i][j] - mu) * rstd * g4[j] + b4[j];
#if defined(MDR_LN_ABL) && MDR_LN_ABL == 2  // measurement build: no fp16 operand either -- only the fp32 residual stream is read and written
                if (out16 && y[0] == 123456.f) *(half4*)(out16 + (size_t)t * H + e) = (half4){0, 0, 0, 0};
#else
                if (out16) {
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)y[j];
                    *(half4*)(out16 + (size_t)t * H + e) = o;
                }
#endif
#if defined(MDR_LN_ABL) && MDR_LN_ABL == 3  // measurement build: no fp32 residual write
                if (out32 && y[0] == 123456.f) *(f32x4*)(out32 + (size_t)t * H + e) = y;
#else
                if (out32) *(f32x4*)(out32 + (size_t)t * H + e) = y;
#endif
            }
        return;
    }
    const int n = H >> 6;
    float x[kMaxPerLane];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i)
        if (i < n) {
            x[i] = (float)r[lane + 64 * i] + (res16 ? (float)res16[(size_t)t * H + lane + 64 * i] : 0.f) + (res32 ? res32[(size_t)t * H + lane + 64 * i] : 0.f);
            s += x[i];
        }
    const float mu = wave_sum(s) / H;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i)
        if (i < n) { float dlt = x[i] - mu; v += dlt * dlt; }
    const float rstd = rsqrtf(wave_sum(v) / H + eps);
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i)
        if (i < n) {
            int e = lane + 64 * i;
            float y = (x[i] - mu) * rstd * g[e] + bta[e];
            if (out16) out16[(size_t)t * H + e] = (_Float16)y;
            if (out32) out32[(size_t)t * H + e] = y;
        }
}

// first token of every sequence -> dense [B, H] fp16
__global__ void gather_cls_kernel(const _Float16* __restrict__ h, const float* __restrict__ h32, const int* __restrict__ cu, int B, int H,
                                  _Float16* __restrict__ out, float* __restrict__ out32) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    int b = i / H, e = i - b * H;
    out[i] = h[(size_t)cu[b] * H + e];
    if (h32) out32[i] = h32[(size_t)cu[b] * H + e];
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (_Float16)in[i];
}

// csrc/mdr_encoder_gemm_quad.inl -- persistent 256x256x64 GEMM on FOUR waves of 128x128 (one wave per SIMD, 512 registers each).
// Included by mdr_encoder.hip inside namespace mdr::{anonymous}, after mdr_encoder_gemm.inl.
//
// Same tile, LDS image, tile walk, K order and epilogue arithmetic as gemm_big_kernel (bit-identical results), different split:
//   * a wave owns 128x128 outputs = 64 accumulator tiles (the 256 AGPRs of the wave's 512 registers), so a K-tile costs the CU
//     4 x (128 + 128) x 128 B = 128 KiB of LDS fragment reads instead of 8 x (128 + 64) x 128 B = 192 KiB;
//   * ONE barrier per K-tile. A K-tile is two phases (k-half 0 | 1) of 64 MFMAs; the fragments of the next phase are read
//     during the current one into the other half of a double register buffer, so slot s has been read completely when phase
//     (T, 1) starts. The barrier there certifies both "slot s is free" (refilled with K-tile T+2 from then on) and "K-tile T+1
//     has landed" (its k-half 0 is read during phase (T, 1));
//   * the 16 DMA pieces a wave owes per K-tile (buffer_load ... lds: rows past M read as zeros, so the per-lane offsets never
//     change and a tile switch is two scalar moves) are issued right behind that barrier, the youngest with sub-phases of lead.
// The K-loop of a tile is ONE generated asm statement with fixed registers (scripts/gen_gemm_quad_asm.py ->
// mdr_encoder_gemm_quad_loop.inc): hipcc cannot allocate 256 accumulator + 128 fragment registers (every source form tried spilled
// 25-500 registers into the loop), and with a single wave per SIMD every non-MFMA instruction has to sit in the shadow of an MFMA,
// which only a fixed instruction order guarantees. The accumulators stay in a[0:255] across statements; the epilogue below reads
// them with v_accvgpr_read. Nothing else in this kernel may touch AGPRs (checked on the disassembly: scripts/check_quad_agprs.py).
#include "mdr_encoder_gemm_quad_loop.inc"

template <int IDX>
__device__ __forceinline__ float quad_acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(IDX));
    return x;
}
template <int MT, int NT>
__device__ __forceinline__ f32x4 quad_acc_tile() {
    constexpr int B = 4 * (8 * MT + NT);
    return (f32x4){quad_acc_read<B>(), quad_acc_read<B + 1>(), quad_acc_read<B + 2>(), quad_acc_read<B + 3>()};
}

// one 16-row block (m-fragment MT) of the wave's 128 columns: bias (+ GELU), through the per-wave LDS scratch, out as full 128-B lines through a
// buffer descriptor (rows past M dropped by its bounds check, which covers the VGPR offset only: o_lane carries the row, o_col the column base); b8 = the bias of the 128 columns, read once per tile
template <int EPI, int MT>
__device__ __forceinline__ void quad_epilogue_rows(char* scr, const f32x4 (&b8)[8], __amdgpu_buffer_rsrc_t out_rsrc, unsigned o_col, unsigned o_lane, unsigned ldo_b,
                                                   int g, int lr) {
    constexpr bool F16OUT = EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16;
#pragma unroll
    for (int hf = 0; hf < (F16OUT ? 1 : 2); ++hf) {
        auto put = [&](auto ntc, int q) __attribute__((always_inline)) {
            constexpr int NT = decltype(ntc)::value;
            f32x4 v = quad_acc_tile<MT, NT>() + b8[NT];
            if (EPI == EPI_BIAS_GELU_F16) v = gelu_erf4(v);
            if constexpr (F16OUT) *(scr_u32x2*)(scr + lr * 256 + (((q * 2 + (g >> 1)) ^ lr) << 4) + (g & 1) * 8) = cvt_pk_half4(v);
            else *(scr_f32x4*)(scr + lr * 256 + (((q * 4 + g) ^ lr) << 4)) = v;
        };
        if constexpr (F16OUT) {
            put(std::integral_constant<int, 0>{}, 0); put(std::integral_constant<int, 1>{}, 1); put(std::integral_constant<int, 2>{}, 2);
            put(std::integral_constant<int, 3>{}, 3); put(std::integral_constant<int, 4>{}, 4); put(std::integral_constant<int, 5>{}, 5);
            put(std::integral_constant<int, 6>{}, 6); put(std::integral_constant<int, 7>{}, 7);
        } else if (hf == 0) {
            put(std::integral_constant<int, 0>{}, 0); put(std::integral_constant<int, 1>{}, 1); put(std::integral_constant<int, 2>{}, 2);
            put(std::integral_constant<int, 3>{}, 3);
        } else {
            put(std::integral_constant<int, 4>{}, 0); put(std::integral_constant<int, 5>{}, 1); put(std::integral_constant<int, 6>{}, 2);
            put(std::integral_constant<int, 7>{}, 3);
        }
        // 4 passes of 4 rows x 256 B: row = 4 pass + (lane >> 4), 16-B chunk lane & 15
        u32x4 val[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = 4 * ps + g;
            val[ps] = *(const scr_u32x4*)(scr + r * 256 + ((lr ^ r) << 4));
        }
#if defined(MDR_QUAD_ABL) && MDR_QUAD_ABL == 2  // measurement build: everything but the global stores
        asm volatile("" ::"v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]));
        continue;
#endif
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) __builtin_amdgcn_raw_buffer_store_b128(val[ps], out_rsrc, o_lane + (unsigned)(MT * 16 + 4 * ps) * ldo_b, o_col + (unsigned)hf * 256u, 0);
        asm volatile("s_nop 2" ::"v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]) : "memory");  // store-data hazard hipcc does not cover for SGPR-offset stores: see gemm_big_kernel
    }
}

template <int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_quad_kernel(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M_cap,
                 const int* __restrict__ M_dev, int N, int K, void* __restrict__ out, int ldo) {
    using C = GemmB2;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // the only LDS of the kernel: starts at address 0 (the asm XORs slot addresses)
    float* lds_bias = (float*)(lds + C::LDS_BYTES);
    if ((unsigned)(size_t)lds != 0u) __builtin_trap();  // the generated K-loop XORs slot addresses: the ring has to start at LDS address 0
    const int M = M_dev ? min(*M_dev, M_cap) : M_cap;
    const int ntn = N / 256;
    const int ntm = gemm_head_row_tiles((M + 255) / 256, ntn, gridDim.x, gridDim.x / 4);  // row tiles of the 256x256 walk; the rows behind them: gemm_tail_tile, below
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
    const int KT = K / BK;  // even and >= 4 (launch_gemm_quad)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;  // 2 (m) x 2 (n) waves, 128 x 128 outputs each
    const int g = lane >> 4, lr = lane & 15;

    for (int i = tid; i < N; i += 256) lds_bias[i] = bias[i];

    // ---- loader: piece c of a K-tile = rows 32 (c & 7) + (tid >> 3) of A (c < 8) or W, 16-B chunk (tid & 7) ^ (row & 7) ----
    const int ld_row = tid >> 3;
    const unsigned ld_chunk = (unsigned)(((tid & 7) ^ (ld_row & 7)) * 16);
    const unsigned offa0 = (unsigned)ld_row * (unsigned)lda * 2u + ld_chunk, offw0 = (unsigned)ld_row * (unsigned)K * 2u + ld_chunk;
    const unsigned rsa = __builtin_amdgcn_readfirstlane(64u * (unsigned)lda), rsw = __builtin_amdgcn_readfirstlane(64u * (unsigned)K);  // 32 rows, bytes
    auto make_srd = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)p;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);  // stride 0: raw buffer, offsets are bytes
        r[2] = __builtin_amdgcn_readfirstlane(bytes);                          // reads at or past it return zeros
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 srda = make_srd(A, ((unsigned)(M - 1) * (unsigned)lda + (unsigned)K) * 2u);
    const u32x4 srdw = make_srd(W, (unsigned)N * (unsigned)K * 2u);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        out, 0, (int)(((unsigned)(M - 1) * (unsigned)ldo + (unsigned)N) * ((EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) ? 2u : 4u)), 0x00020000);
    const unsigned rda = (unsigned)((wr * 128 + lr) * 128 + ((g ^ (lane & 7)) << 4));
    const unsigned rdw = (unsigned)(C::A_BYTES + (wc * 128 + lr) * 128 + ((g ^ (lane & 7)) << 4));
    const unsigned dst0 = (unsigned)(wave * 1024);
    const unsigned iters = (unsigned)(KT - 3);
    char* scr = lds + C::LDS_BYTES + kPersistBiasMax * 4 + wave * 4096;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bias loads, before any DMA is in flight
    for (int tile = 0; tile < n_my; ++tile) {
        int m0, n0, m1, n1;
        tile_origin(tile, m0, n0);
        tile_origin(tile + 1 < n_my ? tile + 1 : tile, m1, n1);  // (behind the last tile the loader runs on over the same rows: surplus, never read)
        const unsigned soffa = __builtin_amdgcn_readfirstlane((unsigned)m0 * (unsigned)lda * 2u);
        const unsigned soffw = __builtin_amdgcn_readfirstlane((unsigned)n0 * (unsigned)K * 2u);
        const unsigned nexta = __builtin_amdgcn_readfirstlane((unsigned)m1 * (unsigned)lda * 2u);
        const unsigned nextw = __builtin_amdgcn_readfirstlane((unsigned)n1 * (unsigned)K * 2u);
        if (tile == 0)
            asm volatile(MDR_QUAD_PROLOGUE_ASM
                         :
                         : [srda] "s"(srda), [srdw] "s"(srdw), [soffa] "s"(soffa), [soffw] "s"(soffw), [rda] "v"(rda), [rdw] "v"(rdw), [offa0] "v"(offa0),
                           [offw0] "v"(offw0), [rsa] "s"(rsa), [rsw] "s"(rsw), [dst0] "s"(dst0)
                         : "memory", "m0", "scc", "s20", "s21", "s24", "s25", MDR_QUAD_CLOBBER_V);
        asm volatile(MDR_QUAD_KLOOP_ASM
                     :
                     : [srda] "s"(srda), [srdw] "s"(srdw), [soffa] "s"(soffa), [soffw] "s"(soffw), [nexta] "s"(nexta), [nextw] "s"(nextw), [iters] "s"(iters),
                       [rda] "v"(rda), [rdw] "v"(rdw), [offa0] "v"(offa0), [offw0] "v"(offw0), [rsa] "s"(rsa), [rsw] "s"(rsw), [dst0] "s"(dst0)
                     : "memory", "m0", "scc", "s20", "s21", "s24", "s25", MDR_QUAD_CLOBBER_V, MDR_QUAD_CLOBBER_A);
        // Epilogue under the loads in flight (K-tile 0 and the first pieces of K-tile 1 of the next tile), through a 4 KiB per-wave LDS
        // scratch so that every store instruction writes full 128-B lines (see gemm_big_kernel): a 16-row block of the wave's 128
        // columns = 16 x 256 B (f16) or two halves of 16 x 64 columns x 4 B (f32); 16-B chunks XOR-swizzled by the row.
        constexpr unsigned OB = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16) ? 2u : 4u;
#if defined(MDR_QUAD_ABL) && MDR_QUAD_ABL == 1  // measurement build: no epilogue (results wrong)
        continue;
#endif
        f32x4 b8[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) b8[nt] = *(const f32x4*)(lds_bias + n0 + wc * 128 + nt * 16 + 4 * g);
        const unsigned ldo_b = (unsigned)ldo * OB;
        const unsigned o_col = __builtin_amdgcn_readfirstlane((unsigned)(n0 + wc * 128) * OB);
        const unsigned o_lane = (unsigned)(m0 + wr * 128 + g) * ldo_b + (unsigned)lr * 16u;
        quad_epilogue_rows<EPI, 0>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 1>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 2>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 3>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 4>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 5>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 6>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
        quad_epilogue_rows<EPI, 7>(scr, b8, out_rsrc, o_col, o_lane, ldo_b, g, lr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus pieces must have landed before the LDS is released
    if (tail_m0 < M) {  // (uniform over the grid) the partial last round, on 128x128 tiles (mdr_encoder_gemm.inl: gemm_tail_tile); plain compiled code, behind every K-loop statement
        const int ttn = N / 128, Tt = ((M - tail_m0 + 127) / 128) * ttn;
        for (int t = (int)blockIdx.x; t < Tt; t += (int)gridDim.x)
            gemm_tail_tile<EPI, 2, 2, 4>(A, lda, W, bias, M, K, out, ldo, tail_m0 + (t / ttn) * 128, (t % ttn) * 128, lds, tid, wave, lane);
    }
}

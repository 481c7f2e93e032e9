// csrc/mdr_encoder_gemm_duo.inl -- persistent 256x128x64 GEMM on four waves of 128x64 with TWO accumulator sets: the epilogue of tile j-1 runs
// inside the K-loop of tile j. Included by mdr_encoder.hip inside namespace mdr::{anonymous}, after mdr_encoder_gemm_quad.inl.
//
// Why: with K = 768 an output tile is 12 K-tiles, and writing it out (AGPR reads, bias, conversion, the LDS transpose that makes full-line
// stores, the stores) costs 30-45 % of the tile's time during which the matrix pipe idles (profiles/r03_quad_gemm_epilogue_ablation.txt) --
// in gemm_big_kernel and gemm_quad_kernel alike. A 128x64 wave tile needs 128 accumulator registers, so the wave's 256 AGPRs hold two
// tiles: the MFMAs of tile j write one set while the same instruction stream drains the other, one 16-row unit per K-tile, in the shadow of
// the MFMAs. The smaller tile moves 1.5 x the L2->LDS bytes per flop of the 256x256 kernels; what pays for it is the THREE-slot LDS ring
// (48 KiB slots): a K-tile's DMA pieces stay in flight across one barrier (counted vmcnt), which lifts the rate the two-slot loops are held
// to (scripts/ubench/gen_duo3_loop.py: 1.25 PFLOP/s on the K-loop alone against 0.97 with two slots and the same tile).
// Everything inside a tile is generated asm with fixed registers (scripts/gen_gemm_duo_asm.py -> mdr_encoder_gemm_duo_loop.inc); this file
// holds the tile walk, the operands and the epilogue of a workgroup's LAST tile (nothing left to hide it behind). Results are bit-identical
// to the other GEMM kernels (same MFMA, same K order, bias added and rounded the same way).
#ifndef MDR_DUO_LOOP_INC  // measurement builds name an ablated variant of the generated file (scripts/gen_gemm_duo_asm.py --abl n)
#define MDR_DUO_LOOP_INC "mdr_encoder_gemm_duo_loop.inc"
#endif
#include MDR_DUO_LOOP_INC

constexpr int kDuoSlot = 49152, kDuoScratch = 3 * kDuoSlot, kDuoLds = kDuoScratch + 4 * 2048;

// the last tile's epilogue: accumulator set S, unit by unit through the same scratch, with the arithmetic of the asm epilogue
template <int EPI, int S, int MT>
__device__ __forceinline__ void duo_tail_rows(char* scr, const float* __restrict__ bias, void* __restrict__ out, int ldo, int M, int mrow, int ncol0, int g, int lr,
                                              int lane) {
    constexpr bool F16OUT = EPI == EPI_BIAS_F16;
    auto tile = [&](auto ntc) __attribute__((always_inline)) {
        constexpr int NT = decltype(ntc)::value;
        constexpr int B = 128 * S + 4 * (4 * MT + NT);
        const f32x4 b4 = *(const f32x4*)(bias + ncol0 + NT * 16 + 4 * g);
        return (f32x4){quad_acc_read<B>(), quad_acc_read<B + 1>(), quad_acc_read<B + 2>(), quad_acc_read<B + 3>()} + b4;
    };
    const int rr = lane >> 3, rc = lane & 7;
    if constexpr (F16OUT) {
        auto put = [&](auto ntc) __attribute__((always_inline)) {
            constexpr int q = decltype(ntc)::value;
            const f32x4 v = tile(ntc);
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
            *(half4*)(scr + lr * 128 + (((q * 2 + (g >> 1)) ^ (lr & 7)) << 4) + (g & 1) * 8) = o;
        };
        put(std::integral_constant<int, 0>{}); put(std::integral_constant<int, 1>{}); put(std::integral_constant<int, 2>{}); put(std::integral_constant<int, 3>{});
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = rr + 8 * h;
            const f32x4 val = *(const f32x4*)(scr + r * 128 + ((rc ^ (r & 7)) << 4));
            if (mrow + r < M) *(f32x4*)((_Float16*)out + (size_t)(mrow + r) * ldo + ncol0 + rc * 8) = val;
        }
    } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            auto put = [&](auto ntc, int q2) __attribute__((always_inline)) {
                const f32x4 v = tile(ntc);
                *(f32x4*)(scr + lr * 128 + (((q2 * 4 + g) ^ (lr & 7)) << 4)) = v;
            };
            if (hf == 0) { put(std::integral_constant<int, 0>{}, 0); put(std::integral_constant<int, 1>{}, 1); }
            else { put(std::integral_constant<int, 2>{}, 0); put(std::integral_constant<int, 3>{}, 1); }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = rr + 8 * h;
                const f32x4 val = *(const f32x4*)(scr + r * 128 + ((rc ^ (r & 7)) << 4));
                if (mrow + r < M) *(f32x4*)((float*)out + (size_t)(mrow + r) * ldo + ncol0 + hf * 32 + rc * 4) = val;
            }
        }
    }
}
template <int EPI, int S>
__device__ __forceinline__ void duo_tail(char* scr, const float* __restrict__ bias, void* __restrict__ out, int ldo, int M, int mrow, int ncol0, int g, int lr, int lane) {
    duo_tail_rows<EPI, S, 0>(scr, bias, out, ldo, M, mrow + 0, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 1>(scr, bias, out, ldo, M, mrow + 16, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 2>(scr, bias, out, ldo, M, mrow + 32, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 3>(scr, bias, out, ldo, M, mrow + 48, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 4>(scr, bias, out, ldo, M, mrow + 64, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 5>(scr, bias, out, ldo, M, mrow + 80, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 6>(scr, bias, out, ldo, M, mrow + 96, ncol0, g, lr, lane);
    duo_tail_rows<EPI, S, 7>(scr, bias, out, ldo, M, mrow + 112, ncol0, g, lr, lane);
}

// EPI: EPI_BIAS_F16 or EPI_BIAS_F32. LONGK: K / 64 >= 18 (the woven part covers the first 12 K-tiles, a rolled loop the rest); else K / 64 == 12.
template <int EPI, bool LONGK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_duo_kernel(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M_cap,
                const int* __restrict__ M_dev, int N, int K, void* __restrict__ out, int ldo) {
    constexpr bool F16OUT = EPI == EPI_BIAS_F16;
    constexpr int OB = F16OUT ? 2 : 4;  // bytes per output element
    extern __shared__ __attribute__((aligned(16))) char lds[];  // the only LDS of the kernel: starts at address 0
    const int M = M_dev ? min(*M_dev, M_cap) : M_cap;
    const int ntn = N / 128, ntm = (M + 255) / 256;
    const long long T_all = (long long)ntm * ntn;  // tile order and XCD ownership: see gemm_persist_kernel
    const int xcd = blockIdx.x & 7;
    const int t_base = (int)(T_all * xcd / 8), local_tiles = (int)(T_all * (xcd + 1) / 8) - t_base;
    const int lb = blockIdx.x >> 3, G = gridDim.x >> 3;
    if (lb >= local_tiles) return;
    const int n_my = (local_tiles - lb + G - 1) / G;
    auto tile_origin = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = t_base + lb + j * G;
        m0 = (t / ntn) * 256;
        n0 = (t % ntn) * 128;
    };
    const int KT = K / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;  // 2 (m) x 2 (n) waves, 128 x 64 outputs each
    const int g = lane >> 4, lr = lane & 15;

    // ---- loader: piece c of a K-tile = rows 32 (c & 7) + (tid >> 3) of A (c < 8) or of W (4 pieces), 16-B chunk (tid & 7) ^ (row & 7) ----
    const int ld_row = tid >> 3;
    const unsigned ld_chunk = (unsigned)(((tid & 7) ^ (ld_row & 7)) * 16);
    const unsigned offa0 = (unsigned)ld_row * (unsigned)lda * 2u + ld_chunk, offw0 = (unsigned)ld_row * (unsigned)K * 2u + ld_chunk;
    const unsigned rsa = __builtin_amdgcn_readfirstlane(64u * (unsigned)lda), rsw = __builtin_amdgcn_readfirstlane(64u * (unsigned)K);  // 32 rows, bytes
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto make_srd = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long a = (unsigned long long)p;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);  // stride 0: raw buffer, offsets are bytes
        r[2] = __builtin_amdgcn_readfirstlane(bytes);                          // reads at or past it return zeros, writes are dropped
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 srda = make_srd(A, ((unsigned)(M - 1) * (unsigned)lda + (unsigned)K) * 2u);
    const u32x4 srdw = make_srd(W, (unsigned)N * (unsigned)K * 2u);
    const u32x4 srdo = make_srd(out, ((unsigned)(M - 1) * (unsigned)ldo + (unsigned)N) * (unsigned)OB);
    const u32x4 srdb = make_srd(bias, (unsigned)N * 4u);
    const unsigned rda = (unsigned)((wr * 128 + lr) * 128 + ((g ^ (lane & 7)) << 4));
    const unsigned rdw = (unsigned)(32768 + (wc * 64 + lr) * 128 + ((g ^ (lane & 7)) << 4));
    const unsigned dst0 = (unsigned)(wave * 1024);
    const unsigned iters = (unsigned)(LONGK ? (KT - 15) / 3 : (KT - 6) / 3);  // rolled 3-step groups of the woven / of the plain block
    const unsigned iters_plain = (unsigned)((KT - 6) / 3);
    // epilogue operands
    const unsigned scr0 = (unsigned)(kDuoScratch + wave * 2048);
    const unsigned vwr = F16OUT ? scr0 + (unsigned)(lr * 128 + ((((g >> 1) ^ (lr & 7))) << 4) + (g & 1) * 8) : scr0 + (unsigned)(lr * 128 + ((g ^ (lr & 7)) << 4));
    const int rr = lane >> 3, rc = lane & 7;
    const unsigned vrd = scr0 + (unsigned)(rr * 128 + ((rc ^ (rr & 7)) << 4));
    const unsigned vst0 = (unsigned)rr * (unsigned)ldo * (unsigned)OB + (unsigned)rc * 16u;
    const unsigned vst1 = vst0 + 8u * (unsigned)ldo * (unsigned)OB;
    const unsigned vboff = (unsigned)(wc * 64 + 4 * g) * 4u;
    const unsigned ldo16 = __builtin_amdgcn_readfirstlane(16u * (unsigned)ldo * (unsigned)OB);
    char* scr = lds + scr0;

    int pm0 = 0, pn0 = 0;  // the previous tile (its epilogue rides in this tile's K-loop)
    for (int tile = 0; tile < n_my; ++tile) {
        int m0, n0, m1, n1;
        tile_origin(tile, m0, n0);
        tile_origin(tile + 1 < n_my ? tile + 1 : tile, m1, n1);  // (behind the last tile the loader runs on over the same rows: surplus, never read)
        const unsigned soffa = __builtin_amdgcn_readfirstlane((unsigned)m0 * (unsigned)lda * 2u);
        const unsigned soffw = __builtin_amdgcn_readfirstlane((unsigned)n0 * (unsigned)K * 2u);
        const unsigned nexta = __builtin_amdgcn_readfirstlane((unsigned)m1 * (unsigned)lda * 2u);
        const unsigned nextw = __builtin_amdgcn_readfirstlane((unsigned)n1 * (unsigned)K * 2u);
        const unsigned sout = __builtin_amdgcn_readfirstlane(((unsigned)(pm0 + wr * 128) * (unsigned)ldo + (unsigned)(pn0 + wc * 64)) * (unsigned)OB);
        const unsigned sbias = __builtin_amdgcn_readfirstlane((unsigned)pn0 * 4u);
#define MDR_DUO_OPERANDS                                                                                                                                  \
    [srda] "s"(srda), [srdw] "s"(srdw), [srdo] "s"(srdo), [srdb] "s"(srdb), [soffa] "s"(soffa), [soffw] "s"(soffw), [nexta] "s"(nexta), [nextw] "s"(nextw), \
        [iters] "s"(iters), [rda] "v"(rda), [rdw] "v"(rdw), [offa0] "v"(offa0), [offw0] "v"(offw0), [rsa] "s"(rsa), [rsw] "s"(rsw), [dst0] "s"(dst0),       \
        [sout] "s"(sout), [sbias] "s"(sbias), [ldo16] "s"(ldo16), [vwr] "v"(vwr), [vrd] "v"(vrd), [vst0] "v"(vst0), [vst1] "v"(vst1), [vboff] "v"(vboff)
#define MDR_DUO_CLOBBERS "memory", "m0", "scc", "s20", "s21", "s24", "s25", "s26", MDR_DUO_CLOBBER_V, MDR_DUO_CLOBBER_A
        if (tile == 0) {
            asm volatile(MDR_DUO_PRO_ASM
                         :
                         : [srda] "s"(srda), [srdw] "s"(srdw), [soffa] "s"(soffa), [soffw] "s"(soffw), [rda] "v"(rda), [rdw] "v"(rdw), [offa0] "v"(offa0),
                           [offw0] "v"(offw0), [rsa] "s"(rsa), [rsw] "s"(rsw), [dst0] "s"(dst0)
                         : "memory", "m0", "scc", "s20", "s21", "s24", "s25", "s26", MDR_DUO_CLOBBER_V);
            asm volatile(MDR_DUO_PLAIN_ASM
                         :
                         : [srda] "s"(srda), [srdw] "s"(srdw), [soffa] "s"(soffa), [soffw] "s"(soffw), [nexta] "s"(nexta), [nextw] "s"(nextw), [iters] "s"(iters_plain),
                           [rda] "v"(rda), [rdw] "v"(rdw), [offa0] "v"(offa0), [offw0] "v"(offw0), [rsa] "s"(rsa), [rsw] "s"(rsw), [dst0] "s"(dst0)
                         : MDR_DUO_CLOBBERS);
        } else if (tile & 1) {  // MFMAs into set 1, the epilogue of set 0
            if constexpr (LONGK) asm volatile(MDR_DUO_KL_F32_S1_ASM : : MDR_DUO_OPERANDS : MDR_DUO_CLOBBERS);
            else if constexpr (F16OUT) asm volatile(MDR_DUO_K12_F16_S1_ASM : : MDR_DUO_OPERANDS : MDR_DUO_CLOBBERS);
            else asm volatile(MDR_DUO_K12_F32_S1_ASM : : MDR_DUO_OPERANDS : MDR_DUO_CLOBBERS);
        } else {
            if constexpr (LONGK) asm volatile(MDR_DUO_KL_F32_S0_ASM : : MDR_DUO_OPERANDS : MDR_DUO_CLOBBERS);
            else if constexpr (F16OUT) asm volatile(MDR_DUO_K12_F16_S0_ASM : : MDR_DUO_OPERANDS : MDR_DUO_CLOBBERS);
            else asm volatile(MDR_DUO_K12_F32_S0_ASM : : MDR_DUO_OPERANDS : MDR_DUO_CLOBBERS);
        }
        pm0 = m0;
        pn0 = n0;
    }
    // the last tile's epilogue, in the open
    const int mrow = pm0 + wr * 128, ncol0 = pn0 + wc * 64;
    if ((n_my - 1) & 1) duo_tail<EPI, 1>(scr, bias, out, ldo, M, mrow, ncol0, g, lr, lane);
    else duo_tail<EPI, 0>(scr, bias, out, ldo, M, mrow, ncol0, g, lr, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus pieces must have landed before the LDS is released
}

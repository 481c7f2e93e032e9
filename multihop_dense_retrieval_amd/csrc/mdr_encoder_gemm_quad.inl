// csrc/mdr_encoder_gemm_quad.inl -- persistent 256x256x64 GEMM on FOUR waves of 128x128 (one wave per SIMD, 512 registers each).
// Included by mdr_encoder.hip inside namespace mdr::{anonymous}, after mdr_encoder_gemm.inl.
//
// Same tile, LDS image, tile walk, K order and epilogue arithmetic as gemm_big_kernel (bit-identical results), different split:
//   * a wave owns 128x128 outputs = 64 accumulator tiles (256 registers, the AGPR half of the wave's 512), so a K-tile costs the CU
//     4 x (128 + 128) x 128 B = 128 KiB of LDS fragment reads instead of 8 x (128 + 64) x 128 B = 192 KiB;
//   * ONE barrier per K-tile. A K-tile is two phases (k-half 0 | 1) of 64 MFMAs; the fragments of the next phase are read
//     during the current one into the other half of a double register buffer, so slot s has been read completely when phase
//     (T, 1) starts. The barrier there certifies both "slot s is free" (refilled with K-tile T+2 from then on) and "K-tile T+1
//     has landed" (its k-half 0 is read during phase (T, 1));
//   * the 16 DMA pieces a wave owes per K-tile are issued early in that window (MDR_QUAD_SCHED), so the youngest has sub-phases
//     of lead before the barrier that waits for it with vmcnt(0).
// Instruction order inside a phase is pinned (one reader / DMA slot behind every MFMA): a single wave per SIMD has nobody to cover
// for it, every non-MFMA instruction has to sit in the shadow of an MFMA.
#ifndef MDR_QUAD_SCHED
#define MDR_QUAD_SCHED 0
#endif

template <int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_quad_kernel(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W, const float* __restrict__ bias, int M_cap,
                 const int* __restrict__ M_dev, int N, int K, void* __restrict__ out, int ldo) {
    using C = GemmB2;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* lds_bias = (float*)(lds + C::LDS_BYTES);
    const int M = M_dev ? min(*M_dev, M_cap) : M_cap;
    const int ntn = N / 256, ntm = (M + 255) / 256;
    const long long T_all = (long long)ntm * ntn;  // tile order and XCD ownership: see gemm_persist_kernel
    const int xcd = blockIdx.x & 7;
    const int t_base = (int)(T_all * xcd / 8), local_tiles = (int)(T_all * (xcd + 1) / 8) - t_base;
    const int lb = blockIdx.x >> 3, G = gridDim.x >> 3;
    if (lb >= local_tiles) return;
    const int n_my = (local_tiles - lb + G - 1) / G;
    auto tile_origin = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = t_base + lb + j * G;
        m0 = (t / ntn) * 256;
        n0 = (t % ntn) * 256;
    };
    const int KT = K / BK;
    const int total = n_my * KT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;  // 2 (m) x 2 (n) waves, 128 x 128 outputs each
    const int g = lane >> 4, lr = lane & 15;

    for (int i = tid; i < N; i += 256) lds_bias[i] = bias[i];

    // ---- loader: piece c of a K-tile = rows 32 (c & 7) + (tid >> 3) of A (c < 8) or W, 16-B chunk (tid & 7) ^ (row & 7) ----
    const int ld_row = tid >> 3;
    const int ld_chunk = ((tid & 7) ^ (ld_row & 7)) * 8;
    unsigned a_off[8];  // element offsets (rows clamped to M - 1)
    unsigned w_off = 0;
    int ld_tile = 0, ld_kt = 0, ld_T = 0;
    auto set_load_tile = [&](int j) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int ar = m0 + 32 * i + ld_row;
            ar = ar < M ? ar : M - 1;
            a_off[i] = (unsigned)ar * (unsigned)lda + (unsigned)ld_chunk;
        }
        w_off = (unsigned)(n0 + ld_row) * (unsigned)K + (unsigned)ld_chunk;
    };
    auto issue_piece = [&](int c) __attribute__((always_inline)) {
        char* slot = lds + (ld_T & 1) * C::STAGE_BYTES;
        const int k0 = ld_kt * BK;
        const _Float16* src = c < 8 ? A + (a_off[c] + (unsigned)k0) : W + (w_off + (unsigned)(32 * (c - 8) * K + k0));
        char* dst = slot + (c < 8 ? 0 : C::A_BYTES) + ((c & 7) * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds(MDR_GPTR(src), MDR_LPTR(dst), 16, 0, 0);
    };
    auto next_ktile = [&]() __attribute__((always_inline)) {
        ++ld_T;
        if (ld_T < total) {
            if (++ld_kt == KT) { ld_kt = 0; ++ld_tile; set_load_tile(ld_tile); }
        }
    };
    // pieces per sub-phase: [0..3] = phase 1 of step T (right behind the barrier that freed the slot), [4..7] = phase 0 of step T+1
#if MDR_QUAD_SCHED == 0
    constexpr int kSched[8] = {3, 3, 3, 3, 2, 2, 0, 0};
#elif MDR_QUAD_SCHED == 1
    constexpr int kSched[8] = {4, 4, 4, 4, 0, 0, 0, 0};
#elif MDR_QUAD_SCHED == 2
    constexpr int kSched[8] = {2, 2, 2, 2, 2, 2, 2, 2};
#else
    constexpr int kSched[8] = {2, 2, 2, 2, 3, 3, 2, 0};
#endif
    constexpr int kFirst = kSched[0] + kSched[1] + kSched[2] + kSched[3];  // pieces of a K-tile issued in the phase behind the barrier

    const int a_rd = (wr * 128 + lr) * 128, w_rd = C::A_BYTES + (wc * 128 + lr) * 128;
    const int sw[2] = {((0 * 4 + g) ^ (lane & 7)) << 4, ((1 * 4 + g) ^ (lane & 7)) << 4};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bias loads, before any DMA is in flight
    set_load_tile(0);
    // prologue: K-tile 0 completely, then what phase 1 of a step -1 would have issued of K-tile 1
#pragma unroll
    for (int c = 0; c < 16; ++c) issue_piece(c);
    next_ktile();
#pragma unroll
    for (int c = 0; c < kFirst; ++c) issue_piece(c);

    // The accumulators are always accumulated onto (a second, "first K-tile" instantiation of the phase with C = 0 makes hipcc shuttle
    // them between register classes at every loop head; the epilogue zeroes what it has read instead), through asm MFMAs that tie them
    // to a class: m-fragments 0-6 (224 registers) to the AGPRs, 7 (32) to the VGPRs -- a class filled to its last register leaves the
    // allocator no room for a single copy and it answers with scratch.
    f32x4 accA[7][8], accV[1][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i < 7) accA[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            else accV[i - 7][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    half8 af[2][8], wf[2][8];

    // one phase = 64 MFMAs on buffer H (4 sub-phases: m-fragments 0-3 | 4-7 x n-fragments 0-3 | 4-7); behind the MFMAs, in their shadow:
    // the 16 fragment reads of the next phase (k-half H^1 of `rd_slot`) in sub-phases 0-2 and this phase's DMA pieces
    auto phase = [&](auto hsel, const char* rd_slot) __attribute__((always_inline)) {
        constexpr int H = decltype(hsel)::value;
        constexpr int Hn = H ^ 1;
        constexpr int pc = H == 1 ? 0 : kFirst;  // phase 1 starts a K-tile's pieces, phase 0 of the next step finishes them
        const int swn = sw[Hn];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mg = j >> 1, ng = j & 1;
            const int np = kSched[(H == 1 ? 0 : 4) + j];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = i >> 2, n = i & 3;
                // (asm with the accumulator tied in the AGPR class: left to itself hipcc spreads the 64 tiles over VGPRs, AGPRs and scratch)
                if (4 * mg + q < 7)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(accA[4 * mg + q][4 * ng + n]) : "v"(wf[H][4 * ng + n]), "v"(af[H][4 * mg + q]));
                else
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(accV[4 * mg + q - 7][4 * ng + n]) : "v"(wf[H][4 * ng + n]), "v"(af[H][4 * mg + q]));
                // the slot behind MFMA i
                if (j == 0 && i < 4) wf[Hn][i] = *(const half8*)(rd_slot + w_rd + i * 16 * 128 + swn);
                if (j == 0 && i >= 4 && i < 8) af[Hn][i - 4] = *(const half8*)(rd_slot + a_rd + (i - 4) * 16 * 128 + swn);
                if (j == 1 && i < 4) wf[Hn][4 + i] = *(const half8*)(rd_slot + w_rd + (4 + i) * 16 * 128 + swn);
                if (j == 1 && i >= 4 && i < 8) af[Hn][i] = *(const half8*)(rd_slot + a_rd + i * 16 * 128 + swn);
                if (i >= 9 && ((i - 9) & 1) == 0 && (i - 9) / 2 < np) {  // behind MFMAs 9, 11, 13, 15
                    int before = 0;
                    for (int jj = 0; jj < j; ++jj) before += kSched[(H == 1 ? 0 : 4) + jj];
                    issue_piece(pc + before + (i - 9) / 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // K-tile 0 landed -> fragments of phase (0, 0)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFirst) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) wf[0][q] = *(const half8*)(lds + w_rd + q * 16 * 128 + sw[0]);
#pragma unroll
    for (int q = 0; q < 8; ++q) af[0][q] = *(const half8*)(lds + a_rd + q * 16 * 128 + sw[0]);

    int kt = 0, tile = 0;
    for (int T = 0; T < total; ++T) {
        const char* slot = lds + (T & 1) * C::STAGE_BYTES;
        const char* other = lds + ((T & 1) ^ 1) * C::STAGE_BYTES;
        // ---- phase (T, 0): k-half 0; reads k-half 1 of this slot; the rest of K-tile T+1's pieces
        phase(std::integral_constant<int, 0>{}, slot);
        // ---- the barrier of the step: every wave has read slot (T & 1) completely and K-tile T+1 has landed in the other one
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        next_ktile();
        // ---- phase (T, 1): k-half 1; reads k-half 0 of K-tile T+1; the first pieces of K-tile T+2 into the slot just freed
        phase(std::integral_constant<int, 1>{}, other);  // (behind the last K-tile the reads and pieces are surplus: stale LDS, the last tile's rows again)
        if (++kt == KT) {
            kt = 0;
            int m0, n0;
            tile_origin(tile, m0, n0);
            ++tile;
            // Epilogue under the loads in flight, through a 4 KiB per-wave LDS scratch so that every store instruction writes full
            // 128-B lines (see gemm_big_kernel): a 16-row block of the wave's 128 columns = 16 x 256 B (f16) or two halves of
            // 16 x 64 columns x 4 B (f32); 16-B chunks XOR-swizzled by the row.
            char* scr = lds + C::LDS_BYTES + kPersistBiasMax * 4 + wave * 4096;
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the asm MFMAs are invisible to the hazard recogniser: results settle before the first read
            constexpr bool F16OUT = EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int mrow = m0 + wr * 128 + mt * 16;
#pragma unroll
                for (int hf = 0; hf < (F16OUT ? 1 : 2); ++hf) {
#pragma unroll
                    for (int q = 0; q < (F16OUT ? 8 : 4); ++q) {
                        const int nt = F16OUT ? q : 4 * hf + q;
                        const int n = n0 + wc * 128 + nt * 16 + 4 * g;
                        const f32x4 b4 = *(const f32x4*)(lds_bias + n);
                        f32x4 v;
                        if (mt < 7) { v = accA[mt][nt] + b4; accA[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                        else { v = accV[mt - 7][nt] + b4; accV[mt - 7][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                        if (EPI == EPI_BIAS_GELU_F16) v = gelu_erf4(v);
                        if constexpr (F16OUT) {
                            half4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                            *(half4*)(scr + lr * 256 + (((q * 2 + (g >> 1)) ^ lr) << 4) + (g & 1) * 8) = o;
                        } else {
                            *(f32x4*)(scr + lr * 256 + (((q * 4 + g) ^ lr) << 4)) = v;
                        }
                    }
                    // 4 passes of 4 rows x 256 B: row = 4 pass + (lane >> 4), 16-B chunk lane & 15
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int r = 4 * ps + g;
                        const f32x4 val = *(const f32x4*)(scr + r * 256 + ((lr ^ r) << 4));
                        const int m = mrow + r;
                        if (m < M) {
                            if constexpr (F16OUT) *(f32x4*)((_Float16*)out + (size_t)m * ldo + n0 + wc * 128 + lr * 8) = val;
                            else *(f32x4*)((float*)out + (size_t)m * ldo + n0 + wc * 128 + hf * 64 + lr * 4) = val;
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus pieces must have landed before the LDS is released
}

// csrc/mdr_mips_generic.inl -- the generic fp32-FMA kernel (any d multiple of 32, k <= 1024): on-device reference and fallback. Included by mdr_mips.hip.
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

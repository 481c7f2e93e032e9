// Micro-benchmark: what one CU can pull over its L2 path with the GEMM's own access pattern, as a function of the bytes in flight.
//   One 512-thread workgroup per CU; XCD x's 32 workgroups share operand panels like gemm_big_kernel's FFN2 walk does (an "A" panel of
//   256 rows by `a_share` workgroups, one of `a_share` "W" panels by every a_share-th workgroup); per K-tile a wave issues 8 pieces of
//   8 rows x 128 B (rows `row_stride` bytes apart), i.e. 64 KiB per workgroup and K-tile, K-tiles walking along the rows.
//   MODE 0: global_load_lds (DMA) into a per-wave ring of DEPTH 1-KiB slots, `s_waitcnt vmcnt(DEPTH-1)` before every issue
//   MODE 1: global_load_dwordx4 into registers, PF K-tiles (8 pieces each) ahead of their use
// Prints GB/s over the chip and bytes per shader clock per CU (s_memtime deltas of wave 0).
// Build: hipcc --offload-arch=gfx950 -O3 l2_lds_path.hip -o l2_lds_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int MODE>
__global__ void __launch_bounds__(512) path_kernel(const char* __restrict__ A, const char* __restrict__ W, int row_stride, int passes, int a_share,
                                                   int panels_per_xcd, unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const char* a_panel = A + (size_t)(xcd * panels_per_xcd + lb / a_share) * 256 * row_stride;
    const char* w_panel = W + (size_t)(a_share == 1 ? lb : lb % a_share) * 256 * row_stride;
    const int KT = row_stride / 128;
    const unsigned row = tid >> 3, chunk = (tid & 7) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
        int slot = 0;
        for (int p = 0; p < passes; ++p)
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const char* src = (c < 4 ? a_panel : w_panel) + (size_t)((c & 3) * 64 + row) * row_stride + kt * 128 + chunk;
                    wait_vm<DEPTH - 1>();
                    __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(lds + (wave * DEPTH + slot) * 1024), 16, 0, 0);
                    slot = slot + 1 == DEPTH ? 0 : slot + 1;
                }
            }
        wait_vm<0>();
        s[0] = *(const float*)(lds + tid * 4);
    } else {
        constexpr int PF = DEPTH >= 8 ? DEPTH / 8 : 1;  // K-tiles in flight
        f32x4 r[PF][8];
        const int total = passes * KT;
#pragma unroll
        for (int q = 0; q < PF; ++q) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                r[q][c] = *(const f32x4*)((c < 4 ? a_panel : w_panel) + (size_t)((c & 3) * 64 + row) * row_stride + (q % KT) * 128 + chunk);
        }
        for (int T = 0; T < total; T += PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
#pragma unroll
                for (int c = 0; c < 8; ++c) s += r[q][c];
                const int kt = (T + q + PF) % KT;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    r[q][c] = *(const f32x4*)((c < 4 ? a_panel : w_panel) + (size_t)((c & 3) * 64 + row) * row_stride + kt * 128 + chunk);
            }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int c = 0; c < 8; ++c) s += r[q][c];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) atomicAdd(cyc, t1 - t0);
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[blockIdx.x] = s[0];
}

template <int DEPTH, int MODE>
void run(const char* A, const char* W, int row_stride, int a_share, int panels_per_xcd, unsigned long long* cyc, float* sink, const char* what) {
    const int KT = row_stride / 128;
    const int passes = (48 * 40 + KT - 1) / KT;  // ~1920 K-tiles = 120 MiB per CU
    const size_t lds_bytes = MODE == 0 ? 8 * DEPTH * 1024 : 2048;
    hipFuncSetAttribute((const void*)path_kernel<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((path_kernel<DEPTH, MODE>), dim3(256), dim3(512), lds_bytes, 0, A, W, row_stride, passes, a_share, panels_per_xcd, cyc, sink);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((path_kernel<DEPTH, MODE>), dim3(256), dim3(512), lds_bytes, 0, A, W, row_stride, passes, a_share, panels_per_xcd, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes_cu = (double)passes * KT * 65536.0;
    printf("%-44s %s in flight %3d KiB/CU: %7.0f GB/s chip, %5.1f B/clk/CU (%.0f clk per 64-KiB K-tile, %.3f ms)\n", what,
           MODE == 0 ? "DMA->LDS " : "load->VGPR", DEPTH * 8, bytes_cu * 256 / ms / 1e6, bytes_cu / ((double)c / 256), (double)c / 256 / (passes * KT), ms);
}

int main() {
    const int row_stride_max = 6144, panels = 11;
    const size_t a_bytes = (size_t)8 * 32 * 256 * row_stride_max, w_bytes = (size_t)32 * 256 * row_stride_max;
    char *A, *W; unsigned long long* cyc; float* sink;
    hipMalloc(&A, a_bytes); hipMalloc(&W, w_bytes); hipMalloc(&cyc, 8); hipMalloc(&sink, 4096);
    hipMemset(A, 0, a_bytes); hipMemset(W, 0, w_bytes);
    for (int rs : {6144, 1536}) {
        for (int share : {3, 1}) {
            char what[96];
            snprintf(what, sizeof what, "row stride %d B, A panel shared by %d WG%s", rs, share, share == 1 ? " (no L2 reuse)" : "");
            const int ppx = share == 1 ? 32 : panels;
            run<1, 0>(A, W, rs, share, ppx, cyc, sink, what);
            run<2, 0>(A, W, rs, share, ppx, cyc, sink, what);
            run<4, 0>(A, W, rs, share, ppx, cyc, sink, what);
            run<8, 0>(A, W, rs, share, ppx, cyc, sink, what);
            run<12, 0>(A, W, rs, share, ppx, cyc, sink, what);
            run<16, 0>(A, W, rs, share, ppx, cyc, sink, what);
            run<8, 1>(A, W, rs, share, ppx, cyc, sink, what);
            run<16, 1>(A, W, rs, share, ppx, cyc, sink, what);
            run<24, 1>(A, W, rs, share, ppx, cyc, sink, what);
        }
    }
    return 0;
}

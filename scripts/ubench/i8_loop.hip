// Micro-benchmark: the int8 screening pass as a hand-scheduled asm skeleton on four waves of 64 queries (gen_i8_loop.py). 5 M rows x 768 int8.
// Build: python gen_i8_loop.py > i8_loop.inc && hipcc --offload-arch=gfx950 -O3 i8_loop.hip -o i8_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = x & 0x3f3f3f3fu;  // small int8 values
    }
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
i8_loop(const char* __restrict__ X, const char* __restrict__ Q, int n_stages, unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, b = blockIdx.x;
    const int nit = ((n_stages - b + G - 1) / G) / 3 * 3;  // multiple of 3 (static ring slots in the unrolled loop)
    const char* xs = X + (size_t)b * 49664;
    const unsigned long long stride = (unsigned long long)G * 49664ull;
    const unsigned rd = (unsigned)(((lane >> 4) & 1) * 12288 + (lane >> 5) * 256 + (lane & 15) * 16);
    const unsigned off0 = (unsigned)(wave * 6 * 1024 + lane * 16), toff = (unsigned)(lane * 4);
    const unsigned dst0 = (unsigned)(wave * 6 * 1024);
    const unsigned l31 = lane & 31, lh = lane >> 5;
    const unsigned qrow0 = wave * 64 + l31, qrow1 = qrow0 + 32;
    const unsigned qoff = (qrow0 >> 4) * 12288 + (qrow0 & 15) * 16 + lh * 256, qoff1 = (qrow1 >> 4) * 12288 + (qrow1 & 15) * 16 + lh * 256;
    float o0, o1; unsigned t0, t1;
    asm volatile(
#include "i8_loop.inc"
        : [o0] "=v"(o0), [o1] "=v"(o1), [t0] "=v"(t0), [t1] "=v"(t1)
        : [X] "s"(xs), [Q] "s"(Q), [stride] "s"(stride), [nit] "s"(nit), [rd] "v"(rd), [off0] "v"(off0), [toff] "v"(toff), [dst0] "s"(dst0), [qoff] "v"(qoff),
          [qoff1] "v"(qoff1), [wave] "s"(wave)
        : "memory", "m0", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29",
#include "i8_loop_clobbers.inc"
    );
    if (tid == 0) atomicAdd(cyc, ((unsigned long long)t1 << 32) | t0);
    if (o0 + o1 == 123.456f) sink[blockIdx.x] = o0;
}
int main() {
    const size_t rows = 5000000, n_sb = (rows + 31) / 32, n_st = (n_sb + 1) / 2;
    const size_t bytes = (n_st + 4 * 256) * 49664;  // the loader runs two of a workgroup's stages past the end
    char *X, *Q; unsigned long long* cyc; float* sink;
    hipMalloc(&X, bytes); hipMalloc(&Q, 16 * 12288); hipMalloc(&cyc, 8); hipMalloc(&sink, 4096);
    fill<<<1024, 256>>>((unsigned*)X, bytes / 4, 1u); fill<<<16, 256>>>((unsigned*)Q, 16 * 12288 / 4, 7u);
    const int lds_bytes = 3 * 49664;
    hipFuncSetAttribute((const void*)i8_loop, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(i8_loop, dim3(256), dim3(256), lds_bytes, 0, X, Q, (int)n_st, cyc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("5 M rows x 768 int8, 256 queries: %.3f ms per pass (%.2f TB/s of the 3.88 GB plane), %.0f clk per 64-row stage and workgroup\n", ms / 5,
               n_st * 49664.0 / (ms / 5 * 1e-3) / 1e12, (double)c / (256.0 * 5) / (n_st / 256.0));
    }
    return 0;
}

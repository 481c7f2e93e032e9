// Micro-benchmark: sustained MFMA rate of the two f16 shapes with register-resident random operands, 8 waves per CU
// (2 per SIMD), independent accumulators, nothing else in the loop. Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) peak(const half8* __restrict__ in, float* __restrict__ out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096]; }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (SHAPE == 816) {  // v_mfma_i32_16x16x64_i8
        i32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (i32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a[i & 3]), __builtin_bit_cast(i32x4, b[i >> 2]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += (float)(acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]);
    } else if (SHAPE == 832) {  // v_mfma_i32_32x32x32_i8
        i32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a[i & 3]), __builtin_bit_cast(i32x4, b[i >> 2]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += (float)acc[i][j];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int WAVES>
void run(const half8* in, float* out, int blocks, const char* name) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((peak<SHAPE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((peak<SHAPE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5.0 * blocks * WAVES * iters * (SHAPE == 16 ? 16 * 16384.0 : SHAPE == 816 ? 16 * 32768.0 : SHAPE == 832 ? 8 * 65536.0 : 8 * 32768.0);
    const double mf = 5.0 * blocks * WAVES * iters * (SHAPE == 16 || SHAPE == 816 ? 16 : 8);
    printf("%-28s %d waves/CU: %8.1f TFLOP/s  (%.2f ms)  = %.1f ns per MFMA per SIMD\n", name, WAVES, flop / ms / 1e9, ms / 5, ms * 1e6 / (mf / (blocks * 4)));
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = zeros ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    half8* in; float* out;
    hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("operands: %s\n", zeros ? "zeros" : "random");
    run<16, 4>(in, out, 256, "16x16x32 f16");
    run<32, 4>(in, out, 256, "32x32x16 f16");
    run<16, 8>(in, out, 256, "16x16x32 f16");
    run<32, 8>(in, out, 256, "32x32x16 f16");
    run<16, 4>(in, out, 256, "16x16x32 f16");
    run<32, 4>(in, out, 256, "32x32x16 f16");
    run<816, 8>(in, out, 256, "16x16x64 i8 (ops)");
    run<832, 8>(in, out, 256, "32x32x32 i8 (ops)");
    run<816, 4>(in, out, 256, "16x16x64 i8 (ops)");
    run<832, 4>(in, out, 256, "32x32x32 i8 (ops)");
    return 0;
}

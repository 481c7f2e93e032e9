// Semantics probe of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950 (the attention kernels' row reductions, rows4_max / rows4_sum):
// prints, per 16-lane row, where the two results of swap(a = lane, b = 100 + lane) come from.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x] = r[0];
    p[threadIdx.x + 64] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    p[threadIdx.x + 128] = q[0];
    p[threadIdx.x + 192] = q[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) printf("permlane16_swap row %d: r0[lane %d] = %u   r1[lane %d] = %u\n", row, row * 16, h[row * 16], row * 16, h[64 + row * 16]);
    for (int row = 0; row < 4; ++row) printf("permlane32_swap row %d: r0[lane %d] = %u   r1[lane %d] = %u\n", row, row * 16, h[128 + row * 16], row * 16, h[192 + row * 16]);
    return 0;
}

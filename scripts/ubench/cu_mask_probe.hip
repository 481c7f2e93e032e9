// scripts/ubench/cu_mask_probe.hip -- where do the workgroups of a kernel land when its stream carries a CU mask (hipExtStreamCreateWithCUMask)?
// Round 5 probe for "CU-partitioned lanes" (DESIGN.md section 9): is bit i of the mask CU i of XCD i % 8, or CU i % 32 of XCD i / 32, and does a
// masked stream keep a 1-workgroup-per-CU persistent kernel off the other lane's CUs?
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void __launch_bounds__(512) where_kernel(unsigned* out, int spin) {
    extern __shared__ char lds[];  // 120 KiB: one workgroup per CU
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        lds[0] = 1;
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}  // hold the CU so that every workgroup of the grid needs its own
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(const char* name, const std::vector<uint32_t>& mask, int n_wgs) {
    hipStream_t st;
    if (mask.empty()) CK(hipStreamCreate(&st));
    else CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    unsigned* out;
    CK(hipMalloc(&out, n_wgs * 8));
    CK(hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(where_kernel, dim3(n_wgs), dim3(512), 120 * 1024, st, out, 2000 /* 20 us at 100 MHz */);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(2 * n_wgs);
    CK(hipMemcpy(h.data(), out, n_wgs * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::map<unsigned, int>> per;  // xcc -> (se, cu) id -> count
    for (int i = 0; i < n_wgs; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    printf("%-34s %4d workgroups, %.1f us (20 us per round of co-resident workgroups):", name, n_wgs, ms * 1e3);
    int total_cus = 0;
    for (auto& x : per) { printf("  xcc%u:%zu CUs", x.first, x.second.size()); total_cus += (int)x.second.size(); }
    printf("  => %d distinct CUs\n", total_cus);
    CK(hipFree(out));
    CK(hipStreamDestroy(st));
    return 0;
}

// the same kernel CAPTURED into a hipGraph on a masked stream, then launched on `launch_masked ? that stream : a plain one`: which CUs does the replay use?
static int run_graph(const char* name, const std::vector<uint32_t>& mask, int n_wgs, bool launch_masked) {
    hipStream_t st, plain;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    CK(hipStreamCreate(&plain));
    unsigned* out;
    CK(hipMalloc(&out, n_wgs * 8));
    CK(hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(where_kernel, dim3(n_wgs), dim3(512), 120 * 1024, st, out, 2000);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipStream_t ls = launch_masked ? st : plain;
    CK(hipGraphLaunch(ge, ls));
    CK(hipStreamSynchronize(ls));
    std::vector<unsigned> h(2 * n_wgs);
    CK(hipMemcpy(h.data(), out, n_wgs * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::map<unsigned, int>> per;
    for (int i = 0; i < n_wgs; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        per[xcc][(hw >> 8) & 0x7FFF]++;
    }
    int total_cus = 0;
    for (auto& x : per) total_cus += (int)x.second.size();
    printf("%-58s %4d workgroups => %d distinct CUs\n", name, n_wgs, total_cus);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(out)); CK(hipStreamDestroy(st)); CK(hipStreamDestroy(plain));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, ncu);
    const int words = (ncu + 31) / 32;
    run("no mask", {}, ncu);
    std::vector<uint32_t> all(words, 0xFFFFFFFFu);
    run("mask = all ones", all, ncu);
    std::vector<uint32_t> first224(words, 0);
    for (int i = 0; i < 224; ++i) first224[i / 32] |= 1u << (i % 32);
    run("mask = bits 0..223", first224, 224);
    run("mask = bits 0..223, 256 wgs", first224, 256);
    std::vector<uint32_t> last32(words, 0);
    for (int i = 224; i < 256; ++i) last32[i / 32] |= 1u << (i % 32);
    run("mask = bits 224..255", last32, 32);
    std::vector<uint32_t> mod8(words, 0);  // every CU whose index % 8 == 7
    for (int i = 0; i < 256; ++i) if (i % 8 == 7) mod8[i / 32] |= 1u << (i % 32);
    run("mask = bits with i % 8 == 7", mod8, 32);
    std::vector<uint32_t> low32(words, 0);
    low32[0] = 0xFFFFFFFFu;
    run("mask = bits 0..31", low32, 32);
    run_graph("graph captured on bits 0..223, launched on that stream", first224, 224, true);
    run_graph("graph captured on bits 0..223, launched on a plain stream", first224, 224, false);
    run_graph("graph captured on bits 224..255, launched on that stream", last32, 32, true);
    {  // mask read back
        hipStream_t st;
        CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)first224.size(), first224.data()));
        std::vector<uint32_t> back(words, 0);
        CK(hipExtStreamGetCUMask(st, (uint32_t)back.size(), back.data()));
        int bits = 0;
        for (auto w : back) bits += __builtin_popcount(w);
        printf("hipExtStreamGetCUMask of the 224-bit stream: %d bits set\n", bits);
        std::vector<uint32_t> b0(words, 0);
        CK(hipExtStreamGetCUMask(nullptr, (uint32_t)b0.size(), b0.data()));
        bits = 0;
        for (auto w : b0) bits += __builtin_popcount(w);
        printf("hipExtStreamGetCUMask of the null stream: %d bits set\n", bits);
        CK(hipStreamDestroy(st));
    }
    return 0;
}

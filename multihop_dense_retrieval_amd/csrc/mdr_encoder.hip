// csrc/mdr_encoder.hip -- RoBERTa encoder forward + CLS projection for gfx950 (MI355X).
//
// Replaces   project(encoder(input_ids, mask)[0][:, 0, :])
//   /root/reference/mdr/retrieval/models/mhop_retriever.py:23-26,40-41   (RobertaRetriever.encode_q)
//   /root/reference/mdr/retrieval/models/retriever.py:186-190            (RobertaCtxEncoder.forward)
// where `encoder` is HuggingFace RobertaModel (transformers 2.11, third party). Numerics follow the
// apex-O1 regime the reference runs under (eval_mhop_retrieval.py:88-89): fp16 GEMM operands with fp32
// accumulation; LayerNorm, softmax and GELU in fp32. See DESIGN.md §4.
//
// Execution is UNPADDED: tokens whose mask is 0 are dropped up front (the CLS embedding does not
// depend on them) and every kernel works on the packed [T, hidden] token
// matrix; T is only known on the device, so grids are sized for batch*seq_len and surplus tiles exit.
//
// Kernels
//   enc_lens / enc_scan / enc_scatter   packing: lengths, cu_seqlens, token -> (row, position id)
//   embed_ln                            word + position + type embedding gather, LayerNorm -> fp16
//   gemm_f16<EPI,WM,STAGES>             C = A[M,K] x W[N,K]^T on v_mfma_f32_16x16x32_f16, (64..256)x128x64 tiles,
//                                       global_load_lds staging (XOR-swizzled via the source address),
//                                       fused bias / bias+GELU / bias+residual epilogues
//   attention<NT>                       per (sequence, head): K and V^T of the sequence staged in LDS once,
//                                       S = QK^T on MFMA, fp32 softmax in registers, O = PV on MFMA
//   layernorm                           fp32 [T,H] -> fp16 (hidden state) or fp32 (final embedding)
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <algorithm>
#include <type_traits>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mdr_common.h"

namespace mdr {
namespace {

#include "mdr_encoder_pack_ln.inl"
#include "mdr_encoder_gemm.inl"
#include "mdr_encoder_gemm_quad.inl"
#include "mdr_encoder_attention.inl"

}  // namespace
}  // namespace mdr

// ======================================================================================================
// host side
// ======================================================================================================
using namespace mdr;

struct mdr_encoder {
    mdr_encoder_config cfg{};
    int device = 0;
    int num_cus = 256;
    std::vector<void*> allocs;
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    struct Layer {
        _Float16 *wqkv, *wo, *w1, *w2;
        float *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    };
    std::vector<Layer> layers;
    _Float16* wproj = nullptr;
    float *bproj = nullptr, *lnp_g = nullptr, *lnp_b = nullptr;
    float fill_hint = 0.f;  // expected (tokens / (batch * seq_len)) of the next forwards; 0 = unknown (2/3 is assumed)
};

#ifndef MDR_CU_LANES
#define MDR_CU_LANES 0  // 1: measurement build with CU-partitioned lanes (include/mdr_hip_measure.h)
#endif

namespace {

struct Workspace {
    int *lens, *cu, *total, *tok_src, *tok_pid, *order;  // order: sequences by length, longest first (the ring attention kernel's walk)
    _Float16 *h16, *qkv, *ctx, *ffn, *cls16;
    float *pre, *clspre, *h32, *cls32;  // h32 / cls32: the fp32 residual stream (residual_fp32 mode only)
    size_t bytes;
};

Workspace carve(const mdr_encoder_config& c, int B, int L, char* base) {
    Workspace w{};
    size_t o = 0;
    const size_t T = (size_t)B * L;
    auto take = [&](size_t n) { size_t at = o; o += align_up(n, 256); return base ? base + at : (char*)nullptr; };
    w.lens = (int*)take((size_t)B * 4);
    w.cu = (int*)take((size_t)(B + 1) * 4);
    w.total = (int*)take(4);
    w.order = (int*)take((size_t)B * 4);
    w.tok_src = (int*)take(T * 4);
    w.tok_pid = (int*)take(T * 4);
    w.h16 = (_Float16*)take(T * c.hidden * 2);
    w.qkv = (_Float16*)take(T * 3 * c.hidden * 2);
    w.ctx = (_Float16*)take(T * c.hidden * 2);
    w.ffn = (_Float16*)take(T * c.ffn * 2);
    w.pre = (float*)take(T * c.hidden * 4);
    w.cls16 = (_Float16*)take((size_t)B * c.hidden * 2);
    w.clspre = (float*)take((size_t)B * c.hidden * 4);
    w.h32 = c.residual_fp32 ? (float*)take(T * c.hidden * 4) : nullptr;
    w.cls32 = c.residual_fp32 ? (float*)take((size_t)B * c.hidden * 4) : nullptr;
    w.bytes = o + 256;
    return w;
}

template <int EPI, typename C>
int launch_gemm_cfg(const _Float16* A, int lda, const _Float16* W, const float* bias, int M_cap, const int* M_dev, int N, int K, void* out, int ldo,
                    const _Float16* res, int ldr, hipStream_t st) {
    { int rc_ = ensure_dynamic_lds((const void*)gemm_f16_kernel<EPI, C>, C::LDS_BYTES); if (rc_) return rc_; }
    const int blocks = (N / C::BN) * ((M_cap + C::BM - 1) / C::BM);
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, C>), dim3(blocks), dim3(C::THREADS), C::LDS_BYTES, st, A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

// Rounds of tiles the busiest workgroup walks (tiles split evenly over 8 XCDs, then round-robin over an XCD's workgroups).
inline int persistent_rounds(int M_est, int bm, int N, int bn, int wgs_per_xcd) {
    const long long T = (long long)((M_est + bm - 1) / bm) * (N / bn);
    const long long per_xcd = (T + 7) / 8;
    return (int)((per_xcd + wgs_per_xcd - 1) / wgs_per_xcd);
}

template <int EPI, typename C>
int launch_gemm_persist(const _Float16* A, int lda, const _Float16* W, const float* bias, int M_cap, const int* M_dev, int N, int K, void* out, int ldo,
                        int M_est, int num_cus, hipStream_t st) {
    constexpr int lds = C::LDS_BYTES + kPersistBiasMax * 4;
    { int rc_ = ensure_dynamic_lds((const void*)gemm_persist_kernel<EPI, C>, lds); if (rc_) return rc_; }
    const int grid = num_cus / 8 * 8;
    // MDR_GEMM_EPI (compile-time, measurement builds): 0 stores after the tile, 1 deferred, 2 (product) deferred + two stores may stay in flight
    hipLaunchKernelGGL((gemm_persist_kernel<EPI, C>), dim3(grid), dim3(C::THREADS), lds, st, A, lda, W, bias, M_cap, M_dev, N, K, out, ldo,
                       MDR_GEMM_EPI);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

template <int EPI>
int launch_gemm_big(const _Float16* A, int lda, const _Float16* W, const float* bias, int M_cap, const int* M_dev, int N, int K, void* out, int ldo,
                    int M_est, int num_cus, hipStream_t st) {
    constexpr int lds = GemmB2::LDS_BYTES + kPersistBiasMax * 4 + 8 * 2048;  // slots + bias + per-wave epilogue scratch
    const int grid = num_cus / 8 * 8;
    { int rc_ = ensure_dynamic_lds((const void*)gemm_big_kernel<EPI>, lds); if (rc_) return rc_; }
    hipLaunchKernelGGL((gemm_big_kernel<EPI>), dim3(grid), dim3(512), lds, st, A, lda, W, bias, M_cap, M_dev, N, K, out, ldo);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

template <int EPI>
int launch_gemm_quad(const _Float16* A, int lda, const _Float16* W, const float* bias, int M_cap, const int* M_dev, int N, int K, void* out, int ldo,
                     int M_est, int num_cus, hipStream_t st) {
    constexpr int lds = GemmB2::LDS_BYTES + kPersistBiasMax * 4 + 4 * 4096;  // slots + bias + per-wave epilogue scratch
    const int grid = num_cus / 8 * 8;
    { int rc_ = ensure_dynamic_lds((const void*)gemm_quad_kernel<EPI>, lds); if (rc_) return rc_; }
    hipLaunchKernelGGL((gemm_quad_kernel<EPI>), dim3(grid), dim3(256), lds, st, A, lda, W, bias, M_cap, M_dev, N, K, out, ldo);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

// M_est: expected number of valid rows (the packed token count is only known on the device).
// *res_added tells the caller whether the residual went into the output (else the following LayerNorm adds it).
template <int EPI>
int launch_gemm(const _Float16* A, int lda, const _Float16* W, const float* bias, int M_cap, const int* M_dev, int N, int K, void* out, int ldo,
                const _Float16* res, int ldr, int M_est, int num_cus, hipStream_t st, bool* res_added = nullptr, int force = -1) {
    // experiment knob: 1 small, 2 mid, 3 big tiles; 4 persistent 256x128 / 6 persistent 256x256 for the large-M calls (the others keep the heuristic)
    static int env_sel = getenv("MDR_GEMM_CFG") ? atoi(getenv("MDR_GEMM_CFG")) : 0;
    int sel = force >= 0 ? force : env_sel;
    if (force < 0 && (sel == 4 || sel == 6 || sel == 7) && (long long)(N / 128) * ((M_est + 255) / 256) < (long long)num_cus * 3 / 2) sel = 0;
    if (res_added) *res_added = true;
    const long long p_tiles = (long long)(N / 128) * ((M_est + 255) / 256);
    if ((sel == 4 || sel == 6 || sel == 7 || (sel == 0 && p_tiles >= (long long)num_cus * 3 / 2)) && N % 128 == 0 && N <= kPersistBiasMax) {
        if (res_added) *res_added = false;
        else if (EPI == EPI_BIAS_RES_F32) return set_error(MDR_E_STATE, "large-M GEMM with a residual needs the caller to take the residual (res_added)");
        constexpr int E = EPI == EPI_BIAS_RES_F32 ? EPI_BIAS_F32 : EPI;
        // Both kernels are bound by the bytes a CU moves over its L2 path, loads AND stores (~20 B/clk/CU measured; skipping
        // the stores made the K = 768 GEMMs 25-30 % faster, deferring them did not): cost = rounds x (tile inputs + outputs).
        // the 256x256 kernels store through a buffer descriptor with 32-bit byte offsets: outputs of 4 GiB and more go to the 256x128 kernel
        // (the offsets of the LAST row tile reach row M + 254: the guard covers M_cap + 255 rows, so that they cannot wrap into valid low rows -- ADVICE r3)
        const bool out32 = ((unsigned long long)M_cap + 255ull) * (unsigned long long)ldo * (E == EPI_BIAS_F32 ? 4ull : 2ull) < (1ull << 32);
        const bool a32 = ((unsigned long long)M_cap + 255ull) * (unsigned long long)lda * 2ull < (1ull << 32);  // the four-wave kernel also LOADS A through a descriptor
        if (sel == 7 && out32 && a32 && N % 256 == 0 && K % 128 == 0 && K >= 256) return launch_gemm_quad<E>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, M_est, num_cus, st);
        if ((sel == 6 || sel == 0) && N % 256 == 0 && out32) {
            const double osz = (E == EPI_BIAS_F32) ? 4.0 : 2.0;
            // the 256x256 kernels walk only the row tiles that fill complete rounds and finish the rest on 128x128 tiles (gemm_head_row_tiles): cost =
            // complete rounds x big tile + passes x small tile (latency-bound: counted 1.5 x its bytes)
            const int grid = num_cus / 8 * 8, ntm_all = (M_est + 255) / 256;
            auto cost256 = [&](int max_rem, int tiles_per_pass, int* rounds_out) {
                const int head = gemm_head_row_tiles(ntm_all, N / 256, grid, max_rem);
                const int rounds = persistent_rounds(std::min(M_est, head * 256), 256, N, 256, num_cus / 8);
                const long long tail_tiles = (long long)((std::max(0, M_est - head * 256) + 127) / 128) * (N / 128);
                const long long passes = (tail_tiles + (long long)grid * tiles_per_pass - 1) / ((long long)grid * tiles_per_pass);
                if (rounds_out) *rounds_out = rounds;
                return rounds * ((256.0 + 256.0) * K * 2 + 256.0 * 256.0 * osz) + passes * 1.2 * ((128.0 + 128.0) * K * 2 + 128.0 * 128.0 * osz);
            };
            int rounds_q = 0;
            const double c_quad = cost256(grid / 4, 1, &rounds_q), c_big = cost256(grid / 2, 1, nullptr);
            const double c_p = persistent_rounds(M_est, 256, N, 128, num_cus / 8) * ((256.0 + 128.0) * K * 2 + 256.0 * 128.0 * osz);
            if (sel == 6 || std::min(c_big, c_quad) < c_p) {
                // four waves of 128x128 (gemm_quad_kernel): a faster K-loop (fewer LDS reads, one barrier per K-tile) behind a slower epilogue (one
                // wave per SIMD has nobody to hide its latencies). Measured at 20.3 k rows: FFN2 77 vs 83-86 us, out-projection 30-33 vs 31-37,
                // but QKV 84-89 vs 70-72 and FFN1 113-123 vs 106-117 (three / four tiles per workgroup, K = 768): taken where the K-loop dominates.
                if (sel == 0 && a32 && K % 128 == 0 && K >= 256 && (rounds_q == 1 || K >= 2048) && c_quad <= 1.15 * c_big)
                    return launch_gemm_quad<E>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, M_est, num_cus, st);
                return launch_gemm_big<E>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, M_est, num_cus, st);
            }
        }
        return launch_gemm_persist<E, GemmP>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, M_est, num_cus, st);
    }
    const long long big_blocks = (N % 256 == 0) ? (long long)(N / 256) * ((M_est + 255) / 256) : 0;
    const long long mid_blocks = (long long)(N / 128) * ((M_est + 127) / 128);
    const long long small_blocks = (long long)(N / 64) * ((M_est + 63) / 64);
    if (sel == 3 && big_blocks > 0) return launch_gemm_cfg<EPI, GemmBig>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
#ifdef MDR_GEMM_EXTRA_CFGS  // measurement builds: more tile shapes behind the test hook's kernel ids 8-11
    if (sel == 8) return launch_gemm_cfg<EPI, GemmCfg<128, 64, 2, 2, 3>>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
    if (sel == 9) return launch_gemm_cfg<EPI, GemmCfg<64, 128, 2, 2, 3>>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
    if (sel == 10) return launch_gemm_cfg<EPI, GemmCfg<128, 128, 2, 2, 3>>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
    if (sel == 11) return launch_gemm_cfg<EPI, GemmCfg<64, 64, 2, 2, 4>>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
    if (sel == 12) return launch_gemm_cfg<EPI, GemmCfg<128, 64, 4, 2, 3>>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
    if (sel == 13) return launch_gemm_cfg<EPI, GemmCfg<64, 64, 4, 2, 3>>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
#endif
    // bytes the busiest CU pulls through its L2 path: rounds of blocks x (BM + BN) rows of K; ties go to the small tile
    // (measured at 2.4 k rows: QKV / FFN1 faster on 128x128, out-projection / FFN2 on 64x64)
    const long long c_mid = (mid_blocks + num_cus - 1) / num_cus * 256, c_small = (small_blocks + num_cus - 1) / num_cus * 128;
    if (sel == 2 || (sel == 0 && N % 128 == 0 && c_mid < c_small))
        return launch_gemm_cfg<EPI, GemmMid>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
    return launch_gemm_cfg<EPI, GemmSmall>(A, lda, W, bias, M_cap, M_dev, N, K, out, ldo, res, ldr, st);
}

template <int NT>
int launch_attention(const _Float16* qkv, const int* cu, int B, int L, int H, int heads, _Float16* ctx, hipStream_t st) {
    constexpr int lds = attention_lds_bytes<NT>();
    { int rc_ = ensure_dynamic_lds((const void*)attention_kernel<NT>, lds); if (rc_) return rc_; }
    dim3 grid(heads, B);
    hipLaunchKernelGGL((attention_kernel<NT>), grid, dim3(512), lds, st, qkv, cu, H, ctx);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

int launch_attention_ring(const _Float16* qkv, const int* cu, const int* order, int B, int L, int H, int heads, _Float16* ctx, hipStream_t st) {
    { int rc_ = ensure_dynamic_lds((const void*)attention_ring_kernel, kRingLds); if (rc_) return rc_; }
    const int nblk = (L + 127) / 128;
    const long long pairs8 = ((long long)B * heads + 7) / 8 * 8;  // pairs rounded up to whole XCD rounds
    hipLaunchKernelGGL(attention_ring_kernel, dim3((unsigned)(pairs8 * nblk)), dim3(512), kRingLds, st, qkv, cu, order, B, heads, nblk, H, ctx);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

template <int NTC>
int launch_attention_stream(const _Float16* qkv, const int* cu, const int* order, int B, int L, int H, int heads, _Float16* ctx, hipStream_t st) {
    if (MDR_ATTN_RING) return launch_attention_ring(qkv, cu, order, B, L, H, heads, ctx, st);  // (the product; MDR_ATTN_RING=0 builds: the streaming kernel)
    constexpr int lds = NTC * 16 * 128 * 2;
    { int rc_ = ensure_dynamic_lds((const void*)attention_stream_kernel<NTC>, lds); if (rc_) return rc_; }
    dim3 grid(heads, B, (L + 127) / 128);
    hipLaunchKernelGGL((attention_stream_kernel<NTC>), grid, dim3(kAttnThreads), lds, st, qkv, cu, H, ctx);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

const mdr_tensor* find_tensor(const mdr_tensor* ts, int n, const std::string& name) {
    for (int i = 0; i < n; ++i)
        if (ts[i].name && name == ts[i].name) return &ts[i];
    return nullptr;
}

}  // namespace

extern "C" {

int mdr_encoder_create(const mdr_encoder_config* cfg, const mdr_tensor* tensors, int n_tensors, int weights_on_device, int device, void* stream,
                       mdr_encoder** out) {
    MDR_REQUIRE(cfg && tensors && out, "NULL argument");
    MDR_REQUIRE(cfg->hidden > 0 && cfg->hidden % 128 == 0 && cfg->hidden <= 1024, "hidden=%d unsupported (multiple of 128, <= 1024)", cfg->hidden);
    MDR_REQUIRE(cfg->heads > 0 && cfg->hidden == cfg->heads * 64, "head dim must be 64 (hidden=%d heads=%d)", cfg->hidden, cfg->heads);
    MDR_REQUIRE(cfg->ffn > 0 && cfg->ffn % 128 == 0, "ffn=%d must be a multiple of 128", cfg->ffn);
    MDR_REQUIRE(cfg->layers > 0 && cfg->vocab > 0 && cfg->max_pos > 2, "bad geometry");
    MDR_REQUIRE(cfg->residual_fp32 >= 0 && cfg->residual_fp32 <= 2, "residual_fp32=%d must be 0, 1 or 2", cfg->residual_fp32);
    int ndev = 0;
    MDR_HIP_TRY(hipGetDeviceCount(&ndev));
    MDR_REQUIRE(device >= 0 && device < ndev, "device %d out of range", device);
    DeviceGuard guard(device);
    hipStream_t st = (hipStream_t)stream;
    mdr_encoder* h = new (std::nothrow) mdr_encoder();
    MDR_REQUIRE(h != nullptr, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    }
    const int H = cfg->hidden, F = cfg->ffn;

    float* staging = nullptr;
    size_t staging_elems = (size_t)cfg->vocab * H;
    if ((size_t)F * H > staging_elems) staging_elems = (size_t)F * H;
    if ((size_t)H * H > staging_elems) staging_elems = (size_t)H * H;
    if ((size_t)cfg->max_pos * H > staging_elems) staging_elems = (size_t)cfg->max_pos * H;
    int rc = MDR_OK;
    auto fail = [&](int code) {
        if (staging) (void)hipFree(staging);
        mdr_encoder_free(h);
        return code;
    };
    if (hipMalloc((void**)&staging, staging_elems * 4) != hipSuccess) return fail(set_error(MDR_E_HIP, "hipMalloc(staging) failed"));

    // fetch `name` (numel checked) into device fp32 memory at dst
    auto fetch32 = [&](const std::string& name, size_t numel, float* dst) -> int {
        const mdr_tensor* t = find_tensor(tensors, n_tensors, name);
        if (!t) return set_error(MDR_E_INVALID, "missing key in state dict: %s", name.c_str());
        if ((size_t)t->numel != numel) return set_error(MDR_E_INVALID, "size mismatch for %s: expected %zu elements, got %lld", name.c_str(), numel, (long long)t->numel);
        MDR_HIP_TRY(hipMemcpyAsync(dst, t->data, numel * 4, weights_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        return MDR_OK;
    };
    auto alloc = [&](size_t bytes, void** p) -> int {
        MDR_HIP_TRY(hipMalloc(p, bytes));
        h->allocs.push_back(*p);
        return MDR_OK;
    };
    auto keep32 = [&](const std::string& name, size_t numel, float** dst) -> int {
        int r = alloc(numel * 4, (void**)dst);
        if (r) return r;
        return fetch32(name, numel, *dst);
    };
    // fp32 source -> fp16 at dst (dst already allocated)
    auto to16 = [&](const std::string& name, size_t numel, _Float16* dst) -> int {
        int r = fetch32(name, numel, staging);
        if (r) return r;
        hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, st, (const float*)staging, dst, (long long)numel);
        MDR_HIP_TRY(hipGetLastError());
        MDR_HIP_TRY(hipStreamSynchronize(st));  // staging is reused
        return MDR_OK;
    };
#define MDR_TRY(expr) do { rc = (expr); if (rc) return fail(rc); } while (0)

    const std::string E = "encoder.embeddings.";
    MDR_TRY(keep32(E + "word_embeddings.weight", (size_t)cfg->vocab * H, &h->word));
    MDR_TRY(keep32(E + "position_embeddings.weight", (size_t)cfg->max_pos * H, &h->pos));
    MDR_TRY(keep32(E + "token_type_embeddings.weight", (size_t)H, &h->type0));  // row 0 of [type_vocab, H]; type_vocab == 1 for RoBERTa
    MDR_TRY(keep32(E + "LayerNorm.weight", H, &h->emb_g));
    MDR_TRY(keep32(E + "LayerNorm.bias", H, &h->emb_b));
    h->layers.resize(cfg->layers);
    for (int i = 0; i < cfg->layers; ++i) {
        mdr_encoder::Layer& Ly = h->layers[i];
        const std::string P = "encoder.encoder.layer." + std::to_string(i) + ".";
        MDR_TRY(alloc((size_t)3 * H * H * 2, (void**)&Ly.wqkv));
        MDR_TRY(alloc((size_t)3 * H * 4, (void**)&Ly.bqkv));
        const char* qkv_names[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            MDR_TRY(to16(P + "attention.self." + qkv_names[j] + ".weight", (size_t)H * H, Ly.wqkv + (size_t)j * H * H));
            MDR_TRY(fetch32(P + "attention.self." + qkv_names[j] + ".bias", H, Ly.bqkv + (size_t)j * H));
        }
        MDR_TRY(alloc((size_t)H * H * 2, (void**)&Ly.wo));
        MDR_TRY(to16(P + "attention.output.dense.weight", (size_t)H * H, Ly.wo));
        MDR_TRY(keep32(P + "attention.output.dense.bias", H, &Ly.bo));
        MDR_TRY(keep32(P + "attention.output.LayerNorm.weight", H, &Ly.ln1_g));
        MDR_TRY(keep32(P + "attention.output.LayerNorm.bias", H, &Ly.ln1_b));
        MDR_TRY(alloc((size_t)F * H * 2, (void**)&Ly.w1));
        MDR_TRY(to16(P + "intermediate.dense.weight", (size_t)F * H, Ly.w1));
        MDR_TRY(keep32(P + "intermediate.dense.bias", F, &Ly.b1));
        MDR_TRY(alloc((size_t)H * F * 2, (void**)&Ly.w2));
        MDR_TRY(to16(P + "output.dense.weight", (size_t)H * F, Ly.w2));
        MDR_TRY(keep32(P + "output.dense.bias", H, &Ly.b2));
        MDR_TRY(keep32(P + "output.LayerNorm.weight", H, &Ly.ln2_g));
        MDR_TRY(keep32(P + "output.LayerNorm.bias", H, &Ly.ln2_b));
    }
    MDR_TRY(alloc((size_t)H * H * 2, (void**)&h->wproj));
    MDR_TRY(to16("project.0.weight", (size_t)H * H, h->wproj));
    MDR_TRY(keep32("project.0.bias", H, &h->bproj));
    MDR_TRY(keep32("project.1.weight", H, &h->lnp_g));
    MDR_TRY(keep32("project.1.bias", H, &h->lnp_b));
#undef MDR_TRY
    if (hipStreamSynchronize(st) != hipSuccess) return fail(set_error(MDR_E_HIP, "stream sync failed after weight upload"));
    (void)hipFree(staging);
    *out = h;
    return MDR_OK;
}

int mdr_test_gemm_f16(const void* A_dev, const void* W_dev, const float* bias_dev, int M, const int* m_dev, int N, int K, void* out_dev, int epilogue,
                      int kernel, int device, void* stream) {
    MDR_REQUIRE(A_dev && W_dev && bias_dev && out_dev, "NULL pointer");
    MDR_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, "bad GEMM shape M=%d N=%d K=%d (N, K multiples of 64)", M, N, K);
    MDR_REQUIRE(epilogue == EPI_BIAS_F16 || epilogue == EPI_BIAS_GELU_F16 || epilogue == EPI_BIAS_F32, "epilogue must be 0, 1 or 3");
#ifdef MDR_GEMM_EXTRA_CFGS
    MDR_REQUIRE(kernel == 0 || kernel == 1 || kernel == 2 || kernel == 4 || kernel == 6 || kernel == 7 || (kernel >= 8 && kernel <= 13), "kernel must be 0, 1, 2, 4, 6, 7 or 8-13");
#else
    MDR_REQUIRE(kernel == 0 || kernel == 1 || kernel == 2 || kernel == 4 || kernel == 6 || kernel == 7, "kernel must be 0, 1, 2, 4, 6 or 7");
#endif
    DeviceGuard guard(device);
    if (!guard.ok) return set_error(MDR_E_HIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    int ncu = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const _Float16* A = (const _Float16*)A_dev;
    const _Float16* W = (const _Float16*)W_dev;
    hipStream_t st = (hipStream_t)stream;
    if (epilogue == EPI_BIAS_F16) return launch_gemm<EPI_BIAS_F16>(A, K, W, bias_dev, M, m_dev, N, K, out_dev, N, nullptr, 0, M, ncu, st, nullptr, kernel);
    if (epilogue == EPI_BIAS_GELU_F16) return launch_gemm<EPI_BIAS_GELU_F16>(A, K, W, bias_dev, M, m_dev, N, K, out_dev, N, nullptr, 0, M, ncu, st, nullptr, kernel);
    return launch_gemm<EPI_BIAS_F32>(A, K, W, bias_dev, M, m_dev, N, K, out_dev, N, nullptr, 0, M, ncu, st, nullptr, kernel);
}

#if MDR_GEMM_ABL == 5  // measurement builds only (include/mdr_hip_measure.h)
int mdr_test_gemm_stamps(unsigned long long* out8_host, int reset) {
    MDR_REQUIRE(out8_host, "NULL pointer");
    MDR_HIP_TRY(hipDeviceSynchronize());
    MDR_HIP_TRY(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_gemm_stamp), 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        MDR_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), z, sizeof(z)));
    }
    return MDR_OK;
}
#endif

#if MDR_ATTN_ABL == 9  // measurement builds only (include/mdr_hip_measure.h)
int mdr_test_attn_stamps(unsigned long long* out_host, int max_wgs) {
    MDR_REQUIRE(out_host && max_wgs > 0 && max_wgs <= kAttnStampWgs, "NULL pointer or max_wgs out of range");
    MDR_HIP_TRY(hipDeviceSynchronize());
    MDR_HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_attn_stamp), (size_t)max_wgs * 8 * sizeof(unsigned long long)));
    return MDR_OK;
}
#endif

#if MDR_CU_LANES  // measurement builds only (include/mdr_hip_measure.h): CU-partitioned lanes, a measured negative of round 5
static std::mutex g_cu_mu;
static std::map<hipStream_t, int> g_stream_cus;  // CUs a stream may use, read at its first un-captured forward
int mdr_stream_create_cu_range(int device, int cu_lo, int cu_hi, void** stream_out) {
    MDR_REQUIRE(stream_out != nullptr, "stream_out is NULL");
    DeviceGuard guard(device);
    if (!guard.ok) return set_error(MDR_E_HIP, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    MDR_HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int n = prop.multiProcessorCount;
    MDR_REQUIRE(cu_lo >= 0 && cu_hi <= n && cu_hi - cu_lo >= 8 && cu_lo % 8 == 0 && cu_hi % 8 == 0, "CU range [%d, %d) must be multiples of 8 inside [0, %d)", cu_lo,
                cu_hi, n);
    std::vector<uint32_t> mask((size_t)(n + 31) / 32, 0u);
    for (int i = cu_lo; i < cu_hi; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t st = nullptr;
    MDR_HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *stream_out = (void*)st;
    return MDR_OK;
}

int mdr_stream_destroy(void* stream) {
    {
        std::lock_guard<std::mutex> lk(g_cu_mu);
        g_stream_cus.erase((hipStream_t)stream);
    }
    if (stream) MDR_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return MDR_OK;
}
#endif

int mdr_encoder_set_fill_hint(mdr_encoder* h, float fill) {
    MDR_REQUIRE(h != nullptr, "encoder handle is NULL");
    MDR_REQUIRE(fill >= 0.f && fill <= 1.f, "fill must be in [0, 1] (0 = unknown)");
    h->fill_hint = fill;
    return MDR_OK;
}

int mdr_encoder_free(mdr_encoder* h) {
    if (!h) return MDR_OK;
    DeviceGuard guard(h->device);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return MDR_OK;
}

size_t mdr_encoder_workspace_bytes(const mdr_encoder* h, int batch, int seq_len) {
    if (!h || batch <= 0 || seq_len <= 0) return 0;
    return carve(h->cfg, batch, seq_len, nullptr).bytes;
}

int mdr_encoder_forward(mdr_encoder* h, const int64_t* ids_dev, const int64_t* mask_dev, int batch, int seq_len, float* out_dev, void* workspace_dev,
                        size_t workspace_bytes, void* stream) {
    MDR_REQUIRE(h != nullptr, "encoder handle is NULL");
    MDR_REQUIRE(batch >= 0 && seq_len > 0, "bad shape batch=%d seq_len=%d", batch, seq_len);
    if (batch == 0) return MDR_OK;
    MDR_REQUIRE(ids_dev && mask_dev && out_dev, "NULL pointer");
    MDR_REQUIRE(seq_len <= 512, "seq_len=%d exceeds 512 (RoBERTa has 514 positions)", seq_len);
    MDR_REQUIRE((long long)batch * seq_len < (1ll << 31), "batch*seq_len overflows int32; split the batch");
    const mdr_encoder_config& c = h->cfg;
    const size_t need = carve(c, batch, seq_len, nullptr).bytes;
    if (!workspace_dev || workspace_bytes < need) return set_error(MDR_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    Workspace w = carve(c, batch, seq_len, base);
    const int B = batch, L = seq_len, H = c.hidden, F = c.ffn;
    const int Tcap = B * L;
    // tile-shape heuristics only: the packed token count is known on the device; the host may pass what it expects
    const int Test = h->fill_hint > 0.f ? std::max(1, (int)(h->fill_hint * (float)Tcap)) : Tcap - Tcap / 3;
    int ncu = h->num_cus;
#if MDR_CU_LANES
    {  // CU-partitioned lanes (measurement build): the persistent GEMMs size their grids for the CUs THIS stream may run on. The mask cannot be read while the
       // stream is capturing, so the count of the warm-up call on the same stream is remembered (erased by mdr_stream_destroy); only MASKED streams are rounded
       // to whole XCD shares
        std::lock_guard<std::mutex> lk(g_cu_mu);
        auto it = g_stream_cus.find(st);
        if (it != g_stream_cus.end()) ncu = it->second;
        else {
            uint32_t mask[32] = {0};
            int n = 0;
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone && hipExtStreamGetCUMask(st, 32, mask) == hipSuccess)
                for (uint32_t w_ : mask) n += __builtin_popcount(w_);
            else (void)hipGetLastError();
            if (n >= 8 && n < h->num_cus) ncu = n / 8 * 8;
            if (cs == hipStreamCaptureStatusNone) g_stream_cus[st] = ncu;
        }
    }
#endif
    const long long* ids = (const long long*)ids_dev;
    const long long* mask = (const long long*)mask_dev;

    hipLaunchKernelGGL(enc_lens_kernel, dim3((B + 3) / 4), dim3(256), 0, st, mask, B, L, w.lens);
    const int* order = (MDR_ATTN_SORT && B <= 1024) ? w.order : nullptr;
    hipLaunchKernelGGL(enc_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)w.lens, B, w.cu, w.total, (int*)order);
    hipLaunchKernelGGL(enc_scatter_kernel, dim3((B + 3) / 4), dim3(256), 0, st, ids, mask, B, L, c.pad_id, (const int*)w.cu, w.tok_src, w.tok_pid);
    // Residual stream. residual_fp32 = 0: LayerNorm outputs live as fp16 only (GEMM operand AND residual). residual_fp32 = 1:
    // the apex-O1 regime of the reference -- LayerNorm outputs stay fp32 (w.h32) for the residual adds, and only the copy
    // that feeds the next Linear is rounded to fp16. In that mode no GEMM epilogue adds the (fp16) residual: every
    // LayerNorm call takes it from w.h32 and refreshes w.h32 in place.
    const bool r32 = c.residual_fp32 != 0;
    const bool p16 = c.residual_fp32 == 2;  // out-projection / FFN2 outputs rounded to fp16 before the residual add (what apex O1's F.linear returns)
    _Float16* pre16 = (_Float16*)w.pre;     // (the fp16 sums live in the fp32 buffer's memory)
    _Float16* clspre16 = (_Float16*)w.clspre;
    hipLaunchKernelGGL(embed_ln_kernel, dim3((Tcap + 3) / 4), dim3(256), 0, st, ids, (const int*)w.tok_src, (const int*)w.tok_pid, (const int*)w.total,
                       (const float*)h->word, (const float*)h->pos, (const float*)h->type0, (const float*)h->emb_g, (const float*)h->emb_b, H, c.vocab,
                       c.max_pos, c.ln_eps, w.h16, w.h32);
    MDR_HIP_TRY(hipGetLastError());
    int rc;
    // y = LayerNorm(gemm_out + residual) for `rows` rows: one place that knows where the residual comes from
    auto post_ln = [&](const float* pre, bool res_in_gemm, const _Float16* res16, float* res32, int rows_cap, const int* rows_dev, const float* g_,
                       const float* b_, _Float16* out16, float* out32) {
        if (p16)
            hipLaunchKernelGGL(layernorm_kernel<_Float16>, dim3((rows_cap + 3) / 4), dim3(256), 0, st, (const _Float16*)pre, (const _Float16*)nullptr,
                               (const float*)res32, rows_cap, rows_dev, H, g_, b_, c.ln_eps, out16, out32);
        else
            hipLaunchKernelGGL(layernorm_kernel<float>, dim3((rows_cap + 3) / 4), dim3(256), 0, st, pre, (const _Float16*)(r32 || res_in_gemm ? nullptr : res16),
                               (const float*)(r32 ? res32 : nullptr), rows_cap, rows_dev, H, g_, b_, c.ln_eps, out16, (float*)(r32 ? out32 : nullptr));
    };
    for (int i = 0; i < c.layers; ++i) {
        const mdr_encoder::Layer& Ly = h->layers[i];
        rc = launch_gemm<EPI_BIAS_F16>(w.h16, H, Ly.wqkv, Ly.bqkv, Tcap, w.total, 3 * H, H, w.qkv, 3 * H, nullptr, 0, Test, ncu, st);
        if (rc) return rc;
        if (i + 1 == c.layers) {
            // ---- last layer: everything after the K/V projection only for the CLS rows ([B, H] instead of [T, H]) ----
            hipLaunchKernelGGL(gather_cls_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, (const _Float16*)w.h16, (const float*)w.h32, (const int*)w.cu, B, H,
                               w.cls16, w.cls32);
            hipLaunchKernelGGL(attention_cls_kernel, dim3(c.heads, B), dim3(64), 0, st, (const _Float16*)w.qkv, (const int*)w.cu, H, w.ctx);
            MDR_HIP_TRY(hipGetLastError());
            bool res_in = true;
            if (p16) rc = launch_gemm<EPI_BIAS_F16>(w.ctx, H, Ly.wo, Ly.bo, B, nullptr, H, H, clspre16, H, nullptr, 0, B, ncu, st);
            else if (r32) rc = launch_gemm<EPI_BIAS_F32>(w.ctx, H, Ly.wo, Ly.bo, B, nullptr, H, H, w.clspre, H, nullptr, 0, B, ncu, st);
            else rc = launch_gemm<EPI_BIAS_RES_F32>(w.ctx, H, Ly.wo, Ly.bo, B, nullptr, H, H, w.clspre, H, w.cls16, H, B, ncu, st, &res_in);
            if (rc) return rc;
            post_ln(w.clspre, res_in, w.cls16, w.cls32, B, nullptr, Ly.ln1_g, Ly.ln1_b, w.cls16, w.cls32);
            rc = launch_gemm<EPI_BIAS_GELU_F16>(w.cls16, H, Ly.w1, Ly.b1, B, nullptr, F, H, w.ffn, F, nullptr, 0, B, ncu, st);
            if (rc) return rc;
            if (p16) rc = launch_gemm<EPI_BIAS_F16>(w.ffn, F, Ly.w2, Ly.b2, B, nullptr, H, F, clspre16, H, nullptr, 0, B, ncu, st);
            else if (r32) rc = launch_gemm<EPI_BIAS_F32>(w.ffn, F, Ly.w2, Ly.b2, B, nullptr, H, F, w.clspre, H, nullptr, 0, B, ncu, st);
            else rc = launch_gemm<EPI_BIAS_RES_F32>(w.ffn, F, Ly.w2, Ly.b2, B, nullptr, H, F, w.clspre, H, w.cls16, H, B, ncu, st, &res_in);
            if (rc) return rc;
            post_ln(w.clspre, res_in, w.cls16, w.cls32, B, nullptr, Ly.ln2_g, Ly.ln2_b, w.cls16, w.cls32);
            MDR_HIP_TRY(hipGetLastError());
            break;
        }
        constexpr int attn_sel = MDR_ATTN_FORCE;  // compile-time (measurement builds): 1 = one-shot kernel, 2 = streaming kernel, 0 (product) = by length
        if (attn_sel == 2 || (attn_sel == 0 && L > 128)) {
            rc = L <= 64 ? launch_attention_stream<4>(w.qkv, w.cu, order, B, L, H, c.heads, w.ctx, st)
                         : launch_attention_stream<16>(w.qkv, w.cu, order, B, L, H, c.heads, w.ctx, st);
        } else if (L <= 128) rc = launch_attention<8>(w.qkv, w.cu, B, L, H, c.heads, w.ctx, st);
        else if (L <= 384) rc = launch_attention<24>(w.qkv, w.cu, B, L, H, c.heads, w.ctx, st);
        else rc = launch_attention<32>(w.qkv, w.cu, B, L, H, c.heads, w.ctx, st);
        if (rc) return rc;
        bool res_in = true;
        if (p16) rc = launch_gemm<EPI_BIAS_F16>(w.ctx, H, Ly.wo, Ly.bo, Tcap, w.total, H, H, pre16, H, nullptr, 0, Test, ncu, st);
        else if (r32) rc = launch_gemm<EPI_BIAS_F32>(w.ctx, H, Ly.wo, Ly.bo, Tcap, w.total, H, H, w.pre, H, nullptr, 0, Test, ncu, st);
        else rc = launch_gemm<EPI_BIAS_RES_F32>(w.ctx, H, Ly.wo, Ly.bo, Tcap, w.total, H, H, w.pre, H, w.h16, H, Test, ncu, st, &res_in);
        if (rc) return rc;
        post_ln(w.pre, res_in, w.h16, w.h32, Tcap, w.total, Ly.ln1_g, Ly.ln1_b, w.h16, w.h32);
        rc = launch_gemm<EPI_BIAS_GELU_F16>(w.h16, H, Ly.w1, Ly.b1, Tcap, w.total, F, H, w.ffn, F, nullptr, 0, Test, ncu, st);
        if (rc) return rc;
        if (p16) rc = launch_gemm<EPI_BIAS_F16>(w.ffn, F, Ly.w2, Ly.b2, Tcap, w.total, H, F, pre16, H, nullptr, 0, Test, ncu, st);
        else if (r32) rc = launch_gemm<EPI_BIAS_F32>(w.ffn, F, Ly.w2, Ly.b2, Tcap, w.total, H, F, w.pre, H, nullptr, 0, Test, ncu, st);
        else rc = launch_gemm<EPI_BIAS_RES_F32>(w.ffn, F, Ly.w2, Ly.b2, Tcap, w.total, H, F, w.pre, H, w.h16, H, Test, ncu, st, &res_in);
        if (rc) return rc;
        post_ln(w.pre, res_in, w.h16, w.h32, Tcap, w.total, Ly.ln2_g, Ly.ln2_b, w.h16, w.h32);
        MDR_HIP_TRY(hipGetLastError());
    }
    rc = launch_gemm<EPI_BIAS_F32>(w.cls16, H, h->wproj, h->bproj, B, nullptr, H, H, w.clspre, H, nullptr, 0, B, ncu, st);
    if (rc) return rc;
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, st, (const float*)w.clspre, (const _Float16*)nullptr, (const float*)nullptr, B,
                       (const int*)nullptr, H, (const float*)h->lnp_g, (const float*)h->lnp_b, c.ln_eps, (_Float16*)nullptr, out_dev);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

}  // extern "C"

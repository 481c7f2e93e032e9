// csrc/mdr_mips.hip -- brute-force maximum-inner-product search for gfx950 (MI355X).
//
// Replaces faiss.IndexFlatIP.{add,search} as the reference uses them
// (/root/reference/scripts/eval/eval_mhop_retrieval.py:121-122,155,179). See DESIGN.md §3-4 (layout, kernels) and NEGATIVE_RESULTS.md §3 (how they got there).
//
// Storage (MDR_STORE_F32X2H). Every fp32 element x is kept as an fp16 pair
//     hi = fp16(x)            lo = fp16((x - hi) * 2^11)          x ~= hi + lo * 2^-11   (22-bit mantissa)
// = 4 bytes per element, the same HBM bytes as fp32, kept as TWO planes (all hi, all lo), each laid out
// in MFMA-operand order so that the HBM->LDS DMA and the LDS->register reads are perfectly linear:
//     plane -> row-block rb (16 rows) -> k-block kb (32 columns) -> 1 KiB fragment block
//     fragment block: lane l = (row & 15) + 16 * ((col & 31) >> 3) holds 8 consecutive columns (16 B)
// which is exactly the A/B operand layout of v_mfma_f32_16x16x32_f16.
//
// Search, k == 1, nq <= 128 per pass ("screen" kernel, the headline path): only the HI plane is
// streamed (2 bytes/element) and scored with ONE fp16 MFMA per k-block,  s_hi = qh.xh.  For every
// (q, x):  |q.x - qh.xh| <= B_q = 1.2e-3 * |q| * max_row|x|   (fp16 rounding of both operands, 2^-10,
// plus fp32 accumulation slack), so a row can only be the winner if  s_hi >= (largest s_hi seen) - 2 B_q.
// Those few rows (a handful per query: the running maxima and near-ties) are appended to a candidate
// list and re-scored exactly from both planes by mips_refine_kernel; if the list overflows (pathological
// all-ties corpora) a flag makes the exact 3-MFMA stream kernel below run instead. Results are identical
// to scoring every row exactly; HBM traffic is half the fp32 matrix.
//
// Search, nq <= 128 per pass ("stream" kernel): one 512-thread workgroup per CU, persistent over
// row-blocks. Wave w keeps the (hi, lo) fragments of queries 16w..16w+15 for ALL of K in registers
// (2 * NKB * 4 VGPRs), the corpus streams HBM -> LDS (global_load_lds, 3-deep ring of 1 row-block
// each) -> every wave's MFMA A operand. Three fp16 MFMAs per (row-block, k-block) give an
// fp32-accurate score:   q.x ~= qh.xh + (qh.xl + ql.xh) * 2^-11     (ql.xl ~ 2^-22 dropped)
// top-1 lives in two registers per lane; top-k (k <= 128) in per-wave candidate lists with a
// running threshold. The [nq, N] score matrix is never materialised.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cfloat>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "mdr_common.h"

namespace mdr {
namespace {

#include "mdr_mips_layout.inl"
#include "mdr_mips_exact.inl"
#include "mdr_mips_screen_fp16.inl"
#include "mdr_mips_generic.inl"
#ifndef MDR_MIPS_GEMMK
#define MDR_MIPS_GEMMK 0  // 1: measurement build that also holds the GEMM-structured beam > 1 main pass (mdr_mips_gemmk.inl; a measured negative of round 5)
#endif
#if MDR_MIPS_GEMMK
#include "mdr_mips_gemmk.inl"
#endif
#include "mdr_mips_screen_i8.inl"
#include "mdr_mips_merge.inl"

}  // namespace
}  // namespace mdr

// ======================================================================================================
// host side
// ======================================================================================================
using namespace mdr;

struct mdr_index {
    int d = 0, nkb = 0, storage = 0, device = 0;
    int num_cus = 256;
    long long ntotal = 0;
    long long cap_rows = 0;  // multiple of 32 (one screen-kernel super-block)
    char* hi = nullptr;      // fragment-tiled planes, cap_rows * d * 2 bytes each
    char* lo = nullptr;
    char* i8 = nullptr;      // int8 screening plane (F32X2H storage, d = 768): cap_rows / 32 super-blocks of i8_sb_bytes(d / 64), see mips_screen8_kernel
    float* centre = nullptr; // [3 d] c | 1/w | w: centre and per-column scale of the int8 plane (col_sum_kernel), (0, 1, 1) until the first add() sets
                             // them, frozen afterwards; [3 d .. 5 d) scratch for the column sums
    bool centre_set = false;
    bool cb = false;         // the int8 tier scores the plane against q - lambda c (mdr_mips_screen_i8.inl "Query split"): decided with the centre (flags[11])
    int xexp = 0;            // F32X2H: the planes hold x * 2^-xexp (see convert_to_frag_kernel); fitted to the data by add(), grown by a rescale
    bool xexp_set = false;
    int* flags = nullptr;    // device ints: [0] range error seen by add(), [1] same for queries (ignored), [2] max row |x|^2 (float bits),
                             // [8], [9], [10] int8 tier: max row scale, max s_r (L1(x8_r)/2 + d/4), max sum|c_i (x_i - c_i)| (float bits);
                             // [11] the centre is large against the rows' spread around it (centre_finish_kernel) -> cb
    void* stage = nullptr;   // device staging for host-sourced add()
    size_t stage_bytes = 0;
    // pipelined host upload (upload_host_rows): two pinned staging buffers, a copy stream, "chunk copied" / "chunk converted" events per slot
    void* pin[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipStream_t copy_st = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    int variant = 0;
    bool compact = false;    // MDR_STORE_F32X2H_COMPACT: no int8 screening plane
    const char* last_kernel = "none";
};

namespace {


size_t plane_bytes_per_row(const mdr_index* h) { return (size_t)h->d * 2; }
float row_unscale(const mdr_index* h) { return ldexpf(1.f, h->xexp); }   // 2^E: stored rows -> the caller's scale
float row_scale_inv(const mdr_index* h) { return ldexpf(1.f, -h->xexp); }  // 2^-E
long long pad32(long long n) { return (n + 31) / 32 * 32; }

// the int8 screening plane exists for the storage / dimension the k = 1 screen path serves; MDR_MIPS_I8=0 (read at index creation) leaves it out
bool wants_i8(const mdr_index* h) {
    static const bool off = getenv("MDR_MIPS_I8") && atoi(getenv("MDR_MIPS_I8")) == 0;
    return !off && !h->compact && h->storage != MDR_STORE_BF16 && h->d == 768;
}

int grow(mdr_index* h, long long need_rows, hipStream_t st) {
    long long need = (need_rows + 255) / 256 * 256;  // whole 256-row tiles (mips_gemmk_kernel reads tiles; a multiple of 32 as every other kernel expects)
    if (need <= h->cap_rows) return MDR_OK;
    long long ncap = need;  // first reservation is exact; later ones grow by 1.5x
    if (h->cap_rows) {
        long long geo = (h->cap_rows + h->cap_rows / 2 + 255) / 256 * 256;
        if (geo > ncap) ncap = geo;
    }
    const size_t nbytes = (size_t)ncap * plane_bytes_per_row(h);
    const size_t used = (size_t)pad32(h->ntotal) * plane_bytes_per_row(h);
    char* planes[2] = {nullptr, nullptr};
    char* old[2] = {h->hi, h->lo};
    const int nplanes = h->storage == MDR_STORE_BF16 ? 1 : 2;
    for (int i = 0; i < nplanes; ++i) {
        MDR_HIP_TRY(hipMalloc((void**)&planes[i], nbytes));
        if (used) MDR_HIP_TRY(hipMemcpyAsync(planes[i], old[i], used, hipMemcpyDeviceToDevice, st));
        MDR_HIP_TRY(hipMemsetAsync(planes[i] + used, 0, nbytes - used, st));
    }
    char* n8 = nullptr;
    if (wants_i8(h) && !h->centre) {
        MDR_HIP_TRY(hipMalloc((void**)&h->centre, (size_t)h->d * 5 * 4));
        float init[3 * 1024];  // d <= 1024: c = 0, 1/w = 1, w = 1 until the first add() measures them
        for (int i = 0; i < 3 * h->d; ++i) init[i] = i < h->d ? 0.f : 1.f;
        MDR_HIP_TRY(hipMemcpy(h->centre, init, (size_t)h->d * 3 * 4, hipMemcpyHostToDevice));
    }
    if (wants_i8(h)) {
        const size_t sbb = i8_sb_bytes(h->d / 64), nb8 = (size_t)(ncap / 32 + 1) * sbb, used8 = (size_t)(pad32(h->ntotal) / 32) * sbb;  // + 1: the wide kernel's stages are super-block pairs
        MDR_HIP_TRY(hipMalloc((void**)&n8, nb8));
        if (used8) MDR_HIP_TRY(hipMemcpyAsync(n8, h->i8, used8, hipMemcpyDeviceToDevice, st));
        MDR_HIP_TRY(hipMemsetAsync(n8 + used8, 0, nb8 - used8, st));
    }
    MDR_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < nplanes; ++i)
        if (old[i]) MDR_HIP_TRY(hipFree(old[i]));
    if (h->i8) MDR_HIP_TRY(hipFree(h->i8));
    h->hi = planes[0];
    h->lo = planes[1];
    h->i8 = n8;
    h->cap_rows = ncap;
    return MDR_OK;
}

template <typename T>
int launch_convert(bool bf, const T* src_dev, long long n_valid, long long n_total, int d, long long row0, char* dst_hi, char* dst_lo, int* flags,
                   float xinv, hipStream_t st) {
    long long threads = n_total * (d / 8);
    if (threads == 0) return MDR_OK;
    long long blocks = (threads + 255) / 256;
    if (bf)
        hipLaunchKernelGGL((convert_to_frag_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, src_dev, n_valid, n_total, d, row0, dst_hi, dst_lo, flags, 1.0f);
    else
        hipLaunchKernelGGL((convert_to_frag_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, src_dev, n_valid, n_total, d, row0, dst_hi, dst_lo, flags, xinv);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

constexpr long long kCentreRows = 65536;  // rows of the first add() the int8 plane's centre is averaged over

template <typename T>
int launch_add(mdr_index* h, const T* src_dev, long long n, long long row0, hipStream_t st) {
    const float xinv = h->storage == MDR_STORE_BF16 ? 1.0f : row_scale_inv(h);
    int rc = launch_convert(h->storage == MDR_STORE_BF16, src_dev, n, n, h->d, row0, h->hi, h->lo, h->flags, xinv, st);
    if (rc) return rc;
    hipLaunchKernelGGL(row_norm2_max_kernel<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, src_dev, n, h->d, h->flags, xinv);
    if (h->i8 && n > 0) {
        if (!h->centre_set) {  // the first rows this index ever sees define the centre of its int8 plane (any fixed vector is correct)
            const long long nc = n < kCentreRows ? n : kCentreRows;
            float* sums = h->centre + 3 * (size_t)h->d;
            MDR_HIP_TRY(hipMemsetAsync(sums, 0, (size_t)h->d * 8, st));
            const unsigned gy = (unsigned)((nc + 3) / 4 < 256 ? (nc + 3) / 4 : 256);
            hipLaunchKernelGGL(col_sum_kernel<T>, dim3((unsigned)((h->d + 63) / 64), gy), dim3(256), 0, st, src_dev, nc, h->d, sums);
            hipLaunchKernelGGL(centre_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)sums, h->d, 1.0f / (float)nc, h->centre, h->flags + 11);
            h->centre_set = true;
        }
        const long long want = (n + 3) / 4, cap = (long long)h->num_cus * 32;  // a wave walks rows r, r + 4 * grid, ...: centre and weights stay in registers
        hipLaunchKernelGGL(convert_to_i8_kernel<T>, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, src_dev, n, h->d, row0, h->i8, h->flags + 8,
                           (const float*)h->centre);
    }
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

__global__ void scale_flag_float_kernel(int* __restrict__ flag, float f) { *flag = __float_as_int(__int_as_float(*flag) * f); }

// Fit the index exponent E (planes hold x * 2^-E) to the rows about to be converted; rows [0, row0) are already stored. Synchronises.
template <typename T>
int fit_exponent(mdr_index* h, const T* src_dev, long long n, long long row0, hipStream_t st) {
    if (h->storage == MDR_STORE_BF16 || n == 0) return MDR_OK;
    MDR_HIP_TRY(hipMemsetAsync(h->flags + 4, 0, sizeof(int), st));
    const long long count = n * (long long)h->d;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_kernel<T>, dim3(blocks), dim3(256), 0, st, src_dev, count, h->flags);
    MDR_HIP_TRY(hipGetLastError());
    int bits = 0;
    MDR_HIP_TRY(hipMemcpyAsync(&bits, h->flags + 4, sizeof(int), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    float m;
    memcpy(&m, &bits, 4);
    if (!(m <= 3.0e38f)) return set_error(MDR_E_RANGE, "add(): a value is non-finite; rows were not added");
    if (m == 0.f) return MDR_OK;
    int e = 0;
    (void)frexpf(m, &e);           // m in [2^(e-1), 2^e)
    const int target = e - 10;     // m * 2^-target in [2^9, 2^10): headroom of 2^5 before fp16 overflows
    if (!h->xexp_set) {
        h->xexp = target;
        h->xexp_set = true;
        return MDR_OK;
    }
    if (e - h->xexp > 15) {        // m * 2^-E >= 2^15 would not fit: grow E and shrink what is stored by the same power of two (exact)
        const float f = ldexpf(1.f, h->xexp - target);
        const long long n_vec8 = (row0 + 15) / 16 * 16 * (long long)h->d / 8;
        if (n_vec8 > 0) {
            hipLaunchKernelGGL(rescale_planes_kernel, dim3((unsigned)((n_vec8 + 255) / 256)), dim3(256), 0, st, h->hi, h->lo, n_vec8, f);
            hipLaunchKernelGGL(scale_flag_float_kernel, dim3(1), dim3(1), 0, st, h->flags + 2, f * f);  // max row |x|^2 in stored units
            MDR_HIP_TRY(hipGetLastError());
        }
        h->xexp = target;
    }
    return MDR_OK;
}

int add_any(mdr_index* h, const void* src_dev, int dtype, long long n, long long row0, hipStream_t st) {
    int rc = MDR_OK;
    switch (dtype) {  // (before anything is written: a non-finite value rejects the rows here)
        case MDR_DT_F32: rc = fit_exponent(h, (const float*)src_dev, n, row0, st); break;
        case MDR_DT_BF16: rc = fit_exponent(h, (const unsigned short*)src_dev, n, row0, st); break;
        case MDR_DT_F16: rc = fit_exponent(h, (const _Float16*)src_dev, n, row0, st); break;
        default: break;
    }
    if (rc) return rc;
    switch (dtype) {
        case MDR_DT_F32: return launch_add(h, (const float*)src_dev, n, row0, st);
        case MDR_DT_BF16: return launch_add(h, (const unsigned short*)src_dev, n, row0, st);
        case MDR_DT_F16: return launch_add(h, (const _Float16*)src_dev, n, row0, st);
        default: return set_error(MDR_E_INVALID, "unknown src_dtype %d", dtype);
    }
}

size_t elem_size(int dtype) { return dtype == MDR_DT_F32 ? 4 : 2; }

bool is_bf16(const mdr_index* h) { return h->storage == MDR_STORE_BF16; }
// MDR_MIPS_WIDE=0 keeps every call on the 128-queries-per-pass kernels (measurement knob)
bool wide_pass(int nq) {
    static const bool off = getenv("MDR_MIPS_WIDE") && atoi(getenv("MDR_MIPS_WIDE")) == 0;
    return !off && nq > kStreamQ;
}
// The GEMM-structured main pass of the beam > 1 search (mdr_mips_gemmk.inl) is a MEASURED NEGATIVE of round 5 (four versions, 2.7-2.9 ms per 256-query pass at
// 5 M rows where mips_screenk32_kernel takes 2.3-2.4; NEGATIVE_RESULTS round 5). The product library does not contain it (round 6); a -DMDR_MIPS_GEMMK=1 build does
// and runs it for test-hook variant 5 (tests/test_mips_gpu.py keeps it honest there: same lists, same bits) or with MDR_MIPS_GEMMK=1 in the environment (A/B runs).
bool gemmk_on(const mdr_index* h) {
#if MDR_MIPS_GEMMK
    static const bool env_on = getenv("MDR_MIPS_GEMMK") && atoi(getenv("MDR_MIPS_GEMMK")) == 1;
    return env_on || h->variant == 5;
#else
    (void)h;
    return false;
#endif
}
// Query groups of a call with more than kWideQ queries on the 32-queries-per-wave kernels (round 6). VERDICT r5 item 4 asked for the ceil(nq / 256) passes to share the
// ceil(nq / 32) waves EVENLY instead of 256 + 256 + ... + rest (nq 300 = 160 + 140 instead of 256 + 44, nq 800 = 224 + 3 x 192 instead of 3 x 256 + 32), on the model that a pass
// costs its active waves down to the HBM floor. Measured (profiles/r06_query_groups_ab.txt, r06_query_groups_k_sweep.txt): results bit-identical, time within +-4 % -- a pass costs
// its HBM time PLUS a per-wave term that is the clock falling under matrix-pipe load (profiles/r06_screenk32_overlap_ablation_and_clock.txt), so moving waves between passes moves
// little. What is left is a k-dependence: nq 800 at 6.25 M bf16 rows, even vs old cut: k 8 / 16: -2 %, k 32: equal, k 64: +2 %, k 100: +4 %; 5 M fp32-accurate rows: k 8 -2.5 %,
// k 100 +2.5 %, k 1 +1 % (with long lists the 32-query remainder is cheapest on the 16-queries-per-wave kernels). Rule: the even cut for 2 <= k <= 32, the old cut otherwise.
// Every buffer the groups index (query fragments in 16-query blocks, bounds, thresholds, best, outputs) is linear in the query number, so a group is just (first query, count)
// with the first a multiple of 32. MDR_MIPS_EVEN_GROUPS: 0 = the old cut always, 1 (default) = the rule, 2 = even + groups of <= 128 queries on the 16-queries-per-wave
// kernels, 3 = the even cut always (A/B runs and tests; same results in every mode).
int even_groups_mode() {
    static const int m = getenv("MDR_MIPS_EVEN_GROUPS") ? atoi(getenv("MDR_MIPS_EVEN_GROUPS")) : 1;
    return m;
}
bool even_cut(int k) {
    const int m = even_groups_mode();
    return m >= 2 || (m == 1 && k >= 2 && k <= 32);
}
struct QGroup { int q0, n; };
int wide_group_count(int nq) { return (nq + kWideQ - 1) / kWideQ; }
QGroup wide_group(int nq, int gi, int k) {
    if (!even_cut(k)) return {gi * kWideQ, nq - gi * kWideQ < kWideQ ? nq - gi * kWideQ : kWideQ};
    const int ng = wide_group_count(nq), waves = (nq + 31) / 32;
    const int base = waves / ng, extra = waves % ng;
    const int w0 = gi * base + (gi < extra ? gi : extra), w1 = w0 + base + (gi < extra ? 1 : 0);
    const int end = w1 * 32 < nq ? w1 * 32 : nq;
    return {w0 * 32, end - w0 * 32};
}
bool stream_kernel_supports(const mdr_index* h, int k) { return !is_bf16(h) && h->d == 768 && k <= 128; }
bool screen_kernel_supports(const mdr_index* h, int k) { return h->d == 768 && k <= 256; }
// sample stages per workgroup of the k > 1 screen (mips_screen_kernel MODE 2): one published maximum per stage
int sample_stages_for(int k) { return k > 128 ? 2 * kSampleStagesK : kSampleStagesK; }

enum Path { PATH_GENERIC = 1, PATH_STREAM = 2, PATH_SCREEN = 3 };

struct SearchPlan {
    int path;
    int G;    // workgroups of the stream / screen kernels
    int Gx;   // workgroups of the exact stream kernel (== G on the stream path; the fallback behind the screen kernels)
    int Gg;   // workgroups of the generic kernel (when its lists are needed)
    bool lists_stream, lists_generic;
    size_t off_qhi, off_qlo, off_bound, off_qscale, off_best, off_gmax, off_scand, off_sctl, off_cand, off_cnt, off_kth, total;
    bool i8;  // the int8 screening tier runs in front of the fp16 screen (k == 1)
    int G8w;  // workgroups of its 32-queries-per-wave kernel
    int G8;   // its workgroups: TWO per CU (24.25 KiB super-blocks: three slots are 73 KiB), one's barrier and epilogue under the other's MFMAs.
              // Measured at 5 M rows, planted queries, whole call: 1 per CU 1.013 ms, 1 per CU with 64-row stages 0.995 ms, 2 per CU 0.907 ms.
    size_t off_q8, off_qab, off_qlam, off_ctl8, off_gstar;
    size_t off_zero, zero_bytes, off_gmax_b;  // the k = 1 screen path's zero block (make_plan)
};

#ifndef MDR_I8W_SLOTS
#define MDR_I8W_SLOTS 3  // the 32-queries-per-wave int8 kernel needs > 128 VGPRs: ONE workgroup per CU; its stages are 64 rows (48.5 KiB), two in flight
#endif
#ifndef MDR_I8_SLOTS
#define MDR_I8_SLOTS 3  // variant-build knob: LDS ring depth of the int8 screen kernels (3: two workgroups per CU, 4-6: one)
#endif
// variant 4 = the screen path WITHOUT the int8 tier (tests and A/B runs)
// (run_screen8 serves ONE group of at most kStreamQ queries; more than that goes to the 32-queries-per-wave kernel, which loops over
// groups of 256 -- or, with MDR_MIPS_WIDE=0, stays on the fp16 screen, which loops over groups of 128)
// MDR_MIPS_I8_CB = 0 / 1 forces the query split of the int8 tier off / on (A/B runs; results are the same either way: any lambda is correct)
bool use_cb(const mdr_index* h) {
    static const int force = getenv("MDR_MIPS_I8_CB") ? atoi(getenv("MDR_MIPS_I8_CB")) : -1;
    return force < 0 ? h->cb : force != 0;
}
bool i8_tier(const mdr_index* h, int path, int nq, int k) {
    return h->i8 != nullptr && h->variant != 4 && path == PATH_SCREEN && k == 1 && nq < 65536 && (nq <= kStreamQ || wide_pass(nq));
}

SearchPlan make_plan(const mdr_index* h, int nq, int k) {
    SearchPlan p{};
    const int v = h->variant;
    if (v == PATH_GENERIC) p.path = PATH_GENERIC;
    else if (v == PATH_STREAM) p.path = stream_kernel_supports(h, k) ? PATH_STREAM : PATH_GENERIC;
    else if (screen_kernel_supports(h, k)) p.path = PATH_SCREEN;  // auto or forced screen
    else if (stream_kernel_supports(h, k)) p.path = PATH_STREAM;
    else p.path = PATH_GENERIC;
    const long long n_rb = (h->ntotal + 15) / 16;
    const long long units = p.path == PATH_SCREEN ? (h->ntotal + 31) / 32 : n_rb;
    p.G = (int)(units < h->num_cus ? (units > 0 ? units : 1) : h->num_cus);
    if (p.G > 1024) p.G = 1024;
    p.Gx = (int)(n_rb < h->num_cus ? (n_rb > 0 ? n_rb : 1) : h->num_cus);
    const long long gg = (long long)h->num_cus * 2;
    p.Gg = (int)(n_rb < gg ? (n_rb > 0 ? n_rb : 1) : gg);
    // which candidate-list workspaces this call can touch (incl. the conditional exact pass behind the screen kernel)
    // (the exact pass behind the screen-k kernels: the MFMA stream kernel up to its k = 128, the generic kernel for bf16 rows and for 128 < k <= 256)
    p.lists_stream = (p.path == PATH_STREAM || (p.path == PATH_SCREEN && !is_bf16(h) && stream_kernel_supports(h, k))) && k > 1;
    p.lists_generic = p.path == PATH_GENERIC || (p.path == PATH_SCREEN && (is_bf16(h) || !stream_kernel_supports(h, k)));
    const bool screenk = p.path == PATH_SCREEN && k > 1;
    const bool frag = p.path != PATH_GENERIC;
    const size_t nq_pad = (size_t)((nq + kWideQ - 1) / kWideQ) * kWideQ;  // covers both group sizes (128 and 256 queries per pass)
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    p.off_qhi = take(frag ? nq_pad * h->d * 2 : 0);
    p.off_qlo = take(frag && !is_bf16(h) ? nq_pad * h->d * 2 : 0);
    p.off_bound = take(frag ? nq_pad * 4 : 0);
    p.off_qscale = take(frag ? nq_pad * 4 : 0);
    // ---- the k = 1 screen path's zero block: everything its tiers expect cleared, contiguous, ONE memset per search call (round 5: seven memset nodes
    // between the kernels of a 0.9 ms call were seven ~1.5 us boundaries): best | gmax | gmax_b | gstar | ctl8 | sctl ----
    p.i8 = i8_tier(h, p.path, nq, k);
    p.off_best = take((size_t)(nq > 0 ? nq : 1) * 8);
    p.off_zero = p.off_best;
    p.off_gmax = take(p.path == PATH_SCREEN ? nq_pad * 4 : 0);
    p.off_gmax_b = take(p.path == PATH_SCREEN && p.i8 ? nq_pad * 4 : 0);  // the fp16 tier's sample maxima when it runs BEHIND the int8 tier (which owns gmax)
    p.off_gstar = take(p.i8 ? nq_pad * 8 : 0);
    p.off_ctl8 = take(p.i8 ? 256 : 0);  // [0] a candidate list of the int8 tier overflowed -> the fp16 screen runs
    // k == 1: one private list per wave; k > 1: the [G][kStreamQ] sample maxima
    {
        const long long per_cu = MDR_I8_SLOTS <= 3 ? 2 : 1;  // workgroups of the int8 kernels per CU (LDS: 3 slots are 73 KiB, 6 are 146 KiB)
        p.G8 = (int)(units < per_cu * h->num_cus ? (units > 0 ? units : 1) : per_cu * h->num_cus);
        p.G8w = (int)((units + 1) / 2 < h->num_cus ? ((units + 1) / 2 > 0 ? (units + 1) / 2 : 1) : h->num_cus);  // the 32-queries-per-wave kernel: one per CU, stages of two super-blocks
    }
    const size_t gl = p.i8 && p.G8 > p.G ? (size_t)p.G8 : (size_t)p.G;  // workgroups that own candidate lists
    p.off_sctl = take(p.path == PATH_SCREEN ? 256 + gl * 8 * 4 : 0);            // [0] overflow flag, [64..] per-wave counts
    p.zero_bytes = p.off_sctl + (p.path == PATH_SCREEN ? 256 : 0) - p.off_zero;  // (the per-wave counts behind the head are written before they are read)
    p.off_scand = take(p.path != PATH_SCREEN ? 0 : (k == 1 ? gl * 8 * kWaveCandCap * 8 : (size_t)p.G * sample_stages_for(k) * kWideQ * 4));
    // the screen-k lists and the lists of its conditional exact pass (which runs after them in stream order) share one region
    size_t lists = p.lists_stream ? (size_t)p.Gx * kStreamQ * kStreamCap : (p.lists_generic ? (size_t)p.Gg * kGenericQ * kGenericCap : 0);
    size_t slots = p.lists_stream ? (size_t)p.Gx * kStreamQ : (p.lists_generic ? (size_t)p.Gg * kGenericQ : 0);
    if (screenk) {
        const size_t qc = wide_pass(nq) ? kWideQ : kStreamQ;
        const size_t l2 = (size_t)p.G * qc * kScreenKCap, s2 = (size_t)p.G * qc;
        lists = lists > l2 ? lists : l2;
        slots = slots > s2 ? slots : s2;
    }
    p.off_cand = take(lists * 8);
    p.off_cnt = take(slots * 4);
    p.off_kth = take(slots * 8);
    p.off_q8 = take(p.i8 ? nq_pad * h->d : 0);
    p.off_qab = take(p.i8 ? nq_pad * 16 : 0);
    p.off_qlam = take(p.i8 ? nq_pad * 4 : 0);  // lambda_q of the query split
    p.total = o + 256;
    return p;
}

// generic kernel + merge over all queries in groups of kGenericQ; every launch is skipped on the device when *run_if == 0
template <bool BF>
int run_generic(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, int k, float* D_dev, long long* I_dev, long long id_offset,
                const int* run_if, hipStream_t st) {
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    u64* kth = (u64*)(ws + p.off_kth);
    const int n_rb = (int)((h->ntotal + 15) / 16);
    for (int q0 = 0; q0 < nq; q0 += kGenericQ) {
        const int nqg = nq - q0 < kGenericQ ? nq - q0 : kGenericQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.Gg * kGenericQ * 4, st));
        MDR_HIP_TRY(hipMemsetAsync(kth, 0, (size_t)p.Gg * kGenericQ * 8, st));
        hipLaunchKernelGGL((mips_generic_kernel<BF>), dim3(p.Gg), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, (long long)h->ntotal, n_rb, h->nkb,
                           q_dev + (size_t)q0 * h->d, nqg, cand, cnt, kth, k, run_if, row_unscale(h));
        hipLaunchKernelGGL(merge_lists_kernel, dim3(nqg), dim3(256), 0, st, (const u64*)cand, (const int*)cnt, (const u64*)kth, p.Gg, kGenericQ, kGenericCap,
                           k, D_dev + (size_t)q0 * k, I_dev + (size_t)q0 * k, id_offset, run_if, (const float*)nullptr);
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

template <bool BF>
int run_screen(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, const char* qhi, u64* best, hipStream_t st,
               const int* run_if = nullptr) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 0, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 1, BF>, (int)lds_bytes);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    unsigned* gmax = (unsigned*)(ws + (run_if && p.i8 ? p.off_gmax_b : p.off_gmax));  // behind the int8 tier: its own maxima (cleared with the zero block)
    u64* scand = (u64*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    int* wave_cnt = sctl + 64;
    const int ngroups = (nq + kStreamQ - 1) / kStreamQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const size_t qgroup_bytes = (size_t)kStreamQ * h->d * 2;
    // (gmax and sctl[0..63] are part of the search call's zero block: cleared once by mdr_index_search)
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
        const char* qg = qhi + gi * qgroup_bytes;
        hipLaunchKernelGGL((mips_screen_kernel<NKB, 0, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gi * kStreamQ), nqg, gi * kStreamQ, gmax + (size_t)gi * kStreamQ, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_screen_kernel<NKB, 1, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gi * kStreamQ), nqg, gi * kStreamQ, gmax + (size_t)gi * kStreamQ, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_refine_kernel<BF>), dim3(p.G * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev,
                           (const u64*)scand, (const int*)wave_cnt, best, row_unscale(h), run_if);
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// k == 1, at most 128 queries, int8 plane present: the int8 tier (sample pass, main pass, exact re-scoring of its candidates)
template <bool CB>
int run_screen8(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, u64* best, hipStream_t st) {
    constexpr int NKB8 = 12, NS = MDR_I8_SLOTS;
    const size_t lds_bytes = NS * (size_t)(2 * NKB8 * kFragBytes + kI8Tail);
    int rc_ = ensure_dynamic_lds((const void*)mips_screen8_kernel<NKB8, 0, NS, CB>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen8_kernel<NKB8, 1, NS, CB>, (int)lds_bytes);
    const float* qlam = (const float*)(ws + p.off_qlam);
    if (rc_) return rc_;
    unsigned* gmax = (unsigned*)(ws + p.off_gmax);
    u64* scand = (u64*)(ws + p.off_scand);
    int* wave_cnt = (int*)(ws + p.off_sctl) + 64;
    int* ctl8 = (int*)(ws + p.off_ctl8);
    char* q8 = ws + p.off_q8;
    f32x4* qab = (f32x4*)(ws + p.off_qab);
    const int n_sb = (int)((h->ntotal + 31) / 32);
    u64* gstar = (u64*)(ws + p.off_gstar);
    // (gmax, gstar and ctl8 are part of the search call's zero block: cleared once by mdr_index_search)
    // (q8 / qab were written by prep_queries_both_kernel, together with the fp16 fragments: mdr_index_search)
    hipLaunchKernelGGL((mips_screen8_kernel<NKB8, 0, NS, CB>), dim3(p.G8), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, (const char*)q8,
                       (const f32x4*)qab, nq, 0, gmax, scand, wave_cnt, ctl8, gstar, (const u64*)best, qlam);
    // the sample pass's best-lower-bound rows, re-scored exactly: a first `known` that is up to 2 B tighter than their lower bounds
    hipLaunchKernelGGL(mips_star8_kernel, dim3((nq + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)gstar, nq, best, row_unscale(h));
    hipLaunchKernelGGL((mips_screen8_kernel<NKB8, 1, NS, CB>), dim3(p.G8), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, (const char*)q8,
                       (const f32x4*)qab, nq, 0, gmax, scand, wave_cnt, ctl8, gstar, (const u64*)best, qlam);
    hipLaunchKernelGGL(mips_star8_kernel, dim3((nq + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)gstar, nq, best, row_unscale(h));
    hipLaunchKernelGGL(mips_refine8_kernel, dim3(p.G8 * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)scand,
                       (const int*)wave_cnt, (const unsigned*)gmax, best, ctl8, kI8RefinePerQuery * nq, (const f32x4*)qab, row_unscale(h));
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

// k == 1, more than 128 queries: the same three launches per group of 256 queries on the 32-queries-per-wave kernel
template <bool BF>
int run_screen32(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, const char* qhi, u64* best, hipStream_t st,
                 const int* run_if = nullptr) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen32_kernel<NKB, 0, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen32_kernel<NKB, 1, BF>, (int)lds_bytes);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    unsigned* gmax = (unsigned*)(ws + (run_if && p.i8 ? p.off_gmax_b : p.off_gmax));  // behind the int8 tier: its own maxima (cleared with the zero block)
    u64* scand = (u64*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    int* wave_cnt = sctl + 64;
    const int ngroups = wide_group_count(nq);
    const int n_sb = (int)((h->ntotal + 31) / 32);
    // (gmax and sctl[0..63] are part of the search call's zero block: cleared once by mdr_index_search)
    for (int gi = 0; gi < ngroups; ++gi) {
        const QGroup gq = wide_group(nq, gi, 1);
        const int nqg = gq.n;
        const char* qg = qhi + (size_t)gq.q0 * h->d * 2;
        hipLaunchKernelGGL((mips_screen32_kernel<NKB, 0, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gq.q0), nqg, gq.q0, gmax + (size_t)gq.q0, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_screen32_kernel<NKB, 1, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg,
                           (const float*)(bound + (size_t)gq.q0), nqg, gq.q0, gmax + (size_t)gq.q0, scand, wave_cnt, sctl, run_if);
        hipLaunchKernelGGL((mips_refine_kernel<BF>), dim3(p.G * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev,
                           (const u64*)scand, (const int*)wave_cnt, best, row_unscale(h), run_if);
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// k == 1, more than 128 queries, int8 plane present: the int8 tier per group of 256 queries
template <bool CB>
int run_screen8w(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, u64* best, hipStream_t st) {
    constexpr int NKB8 = 12, NS = MDR_I8W_SLOTS;
    const size_t lds_bytes = NS * 2 * (size_t)(2 * NKB8 * kFragBytes + kI8Tail);  // a stage of this kernel = two super-blocks
    int rc_ = ensure_dynamic_lds((const void*)mips_screen8w_kernel<NKB8, 0, NS, CB>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen8w_kernel<NKB8, 1, NS, CB>, (int)lds_bytes);
    const float* qlam = (const float*)(ws + p.off_qlam);
    if (rc_) return rc_;
    unsigned* gmax = (unsigned*)(ws + p.off_gmax);
    u64* scand = (u64*)(ws + p.off_scand);
    int* wave_cnt = (int*)(ws + p.off_sctl) + 64;
    int* ctl8 = (int*)(ws + p.off_ctl8);
    char* q8 = ws + p.off_q8;
    f32x4* qab = (f32x4*)(ws + p.off_qab);
    const int ngroups = wide_group_count(nq);
    const int n_sb = (int)((h->ntotal + 31) / 32);
    u64* gstar = (u64*)(ws + p.off_gstar);
    // (gmax, gstar and ctl8 are part of the search call's zero block: cleared once by mdr_index_search)
    // (q8 / qab were written by prep_queries_both_kernel, together with the fp16 fragments: mdr_index_search)
    for (int gi = 0; gi < ngroups; ++gi) {
        const QGroup gq = wide_group(nq, gi, 1);
        const int nqg = gq.n;
        const size_t g0 = (size_t)gq.q0;
        const char* qg = q8 + g0 * h->d;
        if (gi) MDR_HIP_TRY(hipMemsetAsync(ctl8 + 3, 0, sizeof(int), st));  // emitted-candidate total of this group's pass
        hipLaunchKernelGGL((mips_screen8w_kernel<NKB8, 0, NS, CB>), dim3(p.G8w), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, qg,
                           (const f32x4*)(qab + g0), nqg, gq.q0, gmax + g0, scand, wave_cnt, ctl8, gstar + g0, (const u64*)best, qlam + g0);
        hipLaunchKernelGGL(mips_star8_kernel, dim3((nqg + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb,
                           q_dev + g0 * h->d, (const u64*)(gstar + g0), nqg, best + g0, row_unscale(h));
        hipLaunchKernelGGL((mips_screen8w_kernel<NKB8, 1, NS, CB>), dim3(p.G8w), dim3(512), lds_bytes, st, (const char*)h->i8, (long long)h->ntotal, n_sb, qg,
                           (const f32x4*)(qab + g0), nqg, gq.q0, gmax + g0, scand, wave_cnt, ctl8, gstar + g0, (const u64*)best, qlam + g0);
        hipLaunchKernelGGL(mips_star8_kernel, dim3((nqg + 15) / 16), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb,
                           q_dev + g0 * h->d, (const u64*)(gstar + g0), nqg, best + g0, row_unscale(h));
        hipLaunchKernelGGL(mips_refine8_kernel, dim3(p.G8w * 8), dim3(256), 0, st, (const char*)h->hi, (const char*)h->lo, h->nkb, q_dev, (const u64*)scand,
                           (const int*)wave_cnt, (const unsigned*)gmax, best, ctl8, kI8RefinePerQuery * nqg, (const f32x4*)qab, row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// 2 <= k <= 128: sample pass -> k-th of the workgroup maxima -> screen-k kernel -> merge/refine, per query group;
// sctl[0] = overflow flag for the conditional exact pass
template <bool BF>
int run_screenk(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, int k, const char* qhi, float* D_dev, long long* I_dev,
                long long id_offset, hipStream_t st) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    const size_t merge_lds = (size_t)kMergeKLds * 8;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 2, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screenk_kernel<NKB, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)merge_screenk_kernel<BF>, (int)merge_lds);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    float* tau0 = (float*)(ws + p.off_gmax);
    unsigned* wgmax = (unsigned*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    const int ngroups = (nq + kStreamQ - 1) / kStreamQ;
    const int nq_pad = ngroups * kStreamQ;
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const int stages = sample_stages_for(k);
    const size_t qgroup_bytes = (size_t)kStreamQ * h->d * 2;
    MDR_HIP_TRY(hipMemsetAsync(sctl, 0, 256, st));
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
        const char* qg = qhi + gi * qgroup_bytes;
        const float* bg = bound + (size_t)gi * kStreamQ;
        float* tg = tau0 + (size_t)gi * kStreamQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.G * kStreamQ * 4, st));
        MDR_HIP_TRY(hipMemsetAsync(wgmax, 0, (size_t)p.G * stages * kStreamQ * 4, st));
        hipLaunchKernelGGL((mips_screen_kernel<NKB, 2, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg, nqg,
                           gi * kStreamQ, wgmax, (u64*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, stages);
        hipLaunchKernelGGL(kth_of_maxima_kernel, dim3(nqg), dim3(256), 0, st, (const unsigned*)wgmax, p.G * stages, k, tg, kStreamQ);
        hipLaunchKernelGGL((mips_screenk_kernel<NKB, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg,
                           (const float*)tg, nqg, cand, cnt, k, sctl);
        hipLaunchKernelGGL((merge_screenk_kernel<BF>), dim3(nqg), dim3(256), merge_lds, st, (const u64*)cand, (const int*)cnt, p.G, k, bg, (const char*)h->hi,
                           (const char*)h->lo, h->nkb, q_dev + (size_t)gi * kStreamQ * h->d, D_dev + (size_t)gi * kStreamQ * k,
                           I_dev + (size_t)gi * kStreamQ * k, id_offset, sctl, kStreamQ, row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

// 2 <= k <= 256 with more than 128 queries: groups of 256 on the 32-queries-per-wave kernels. A LAST group of at most 128 queries (nq = 800: 3 x 256 + 32)
// goes through the 16-queries-per-wave kernels instead (round 4): a pass of theirs is HBM-bound on the hi plane (1.6 ms at 5 M rows), a 32-queries-per-wave
// pass costs its MFMA skeleton whatever the number of queries (2.15 ms). Query fragments, bounds, lists and outputs have the same layout for both.
template <bool BF>
int run_screenk32(mdr_index* h, const SearchPlan& p, char* ws, const float* q_dev, int nq, int k, const char* qhi, float* D_dev, long long* I_dev,
                  long long id_offset, hipStream_t st) {
    constexpr int NKB = 24;
    const size_t lds_bytes = 3 * (size_t)NKB * 2 * kFragBytes;
    const size_t merge_lds = (size_t)kMergeKLds * 8;
    int rc_ = ensure_dynamic_lds((const void*)mips_screen32_kernel<NKB, 2, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screenk32_kernel<NKB, BF>, (int)lds_bytes);
#if MDR_MIPS_GEMMK
    constexpr size_t gemmk_lds = (4 + 4) * 16 * kFragBytes + kWideQ * 8;  // four row-fragment + four query-fragment slots of 16 KiB, list counters, thresholds
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_gemmk_kernel<BF>, (int)gemmk_lds);
#endif
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screen_kernel<NKB, 2, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)mips_screenk_kernel<NKB, BF>, (int)lds_bytes);
    if (!rc_) rc_ = ensure_dynamic_lds((const void*)merge_screenk_kernel<BF>, (int)merge_lds);
    if (rc_) return rc_;
    float* bound = (float*)(ws + p.off_bound);
    float* tau0 = (float*)(ws + p.off_gmax);
    unsigned* wgmax = (unsigned*)(ws + p.off_scand);
    int* sctl = (int*)(ws + p.off_sctl);
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    const int ngroups = wide_group_count(nq);
    const int n_sb = (int)((h->ntotal + 31) / 32);
    const int stages = sample_stages_for(k);
    MDR_HIP_TRY(hipMemsetAsync(sctl, 0, 256, st));
    for (int gi = 0; gi < ngroups; ++gi) {
        const QGroup gq = wide_group(nq, gi, k);
        const int nqg = gq.n;
        const size_t g0 = (size_t)gq.q0;
        const char* qg = qhi + g0 * h->d * 2;
        const float* bg = bound + g0;
        float* tg = tau0 + g0;
        // a group of at most 128 queries takes the 16-queries-per-wave kernels (list stride kStreamQ): under the old cut its remainder group, under the even cut only in mode 2
        const bool narrow = ngroups > 1 && nqg <= kStreamQ && (even_groups_mode() == 2 || (!even_cut(k) && gi == ngroups - 1));
        const int qcap = narrow ? kStreamQ : kWideQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.G * qcap * 4, st));
        MDR_HIP_TRY(hipMemsetAsync(wgmax, 0, (size_t)p.G * stages * qcap * 4, st));
        if (narrow) {
            hipLaunchKernelGGL((mips_screen_kernel<NKB, 2, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg, nqg,
                               gq.q0, wgmax, (u64*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, stages);
            hipLaunchKernelGGL(kth_of_maxima_kernel, dim3(nqg), dim3(256), 0, st, (const unsigned*)wgmax, p.G * stages, k, tg, qcap);
            hipLaunchKernelGGL((mips_screenk_kernel<NKB, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg,
                               (const float*)tg, nqg, cand, cnt, k, sctl);
        } else {
            hipLaunchKernelGGL((mips_screen32_kernel<NKB, 2, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg, nqg,
                               gq.q0, wgmax, (u64*)nullptr, (int*)nullptr, (int*)nullptr, (const int*)nullptr, stages);
            hipLaunchKernelGGL(kth_of_maxima_kernel, dim3(nqg), dim3(256), 0, st, (const unsigned*)wgmax, p.G * stages, k, tg, qcap);
#if MDR_MIPS_GEMMK
            if (gemmk_on(h) && h->ntotal >= 256ll * p.G)  // measurement build: the main pass as a 256 x 256 x 64 GEMM with the screen as its epilogue (mdr_mips_gemmk.inl)
                hipLaunchKernelGGL((mips_gemmk_kernel<BF>), dim3(p.G), dim3(512), gemmk_lds, st, (const char*)h->hi, (long long)h->ntotal, qg, bg, (const float*)tg, nqg,
                                   cand, cnt, sctl);
            else
#endif
                hipLaunchKernelGGL((mips_screenk32_kernel<NKB, BF>), dim3(p.G), dim3(512), lds_bytes, st, (const char*)h->hi, (long long)h->ntotal, n_sb, qg, bg,
                                   (const float*)tg, nqg, cand, cnt, k, sctl);
        }
        hipLaunchKernelGGL((merge_screenk_kernel<BF>), dim3(nqg), dim3(256), merge_lds, st, (const u64*)cand, (const int*)cnt, p.G, k, bg, (const char*)h->hi,
                           (const char*)h->lo, h->nkb, q_dev + g0 * h->d, D_dev + g0 * k,
                           I_dev + g0 * k, id_offset, sctl, qcap, row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

}  // namespace

extern "C" {

int mdr_index_create(int d, int storage, int device, mdr_index** out) {
    MDR_REQUIRE(out != nullptr, "out is NULL");
    MDR_REQUIRE(d > 0 && d % 32 == 0 && d <= 1024, "d=%d unsupported: must be a multiple of 32, <= 1024", d);
    MDR_REQUIRE(storage == MDR_STORE_F32X2H || storage == MDR_STORE_BF16 || storage == MDR_STORE_F32X2H_COMPACT, "unknown storage %d", storage);
    int ndev = 0;
    MDR_HIP_TRY(hipGetDeviceCount(&ndev));
    MDR_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return set_error(MDR_E_HIP, "hipSetDevice(%d) failed", device);
    mdr_index* h = new (std::nothrow) mdr_index();
    MDR_REQUIRE(h != nullptr, "out of host memory");
    h->d = d;
    h->nkb = d / 32;
    h->storage = storage == MDR_STORE_F32X2H_COMPACT ? MDR_STORE_F32X2H : storage;
    h->compact = storage == MDR_STORE_F32X2H_COMPACT;
    h->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    if (hipMalloc((void**)&h->flags, 256) != hipSuccess || hipMemset(h->flags, 0, 256) != hipSuccess) {
        delete h;
        return set_error(MDR_E_HIP, "hipMalloc(flags) failed");
    }
    *out = h;
    return MDR_OK;
}

int mdr_index_free(mdr_index* h) {
    if (!h) return MDR_OK;
    DeviceGuard g(h->device);
    if (h->hi) (void)hipFree(h->hi);
    if (h->lo) (void)hipFree(h->lo);
    if (h->i8) (void)hipFree(h->i8);
    if (h->centre) (void)hipFree(h->centre);
    if (h->flags) (void)hipFree(h->flags);
    if (h->stage) (void)hipFree(h->stage);
    for (int i = 0; i < 2; ++i) {
        if (h->pin[i]) (void)hipHostFree(h->pin[i]);
        if (h->ev_copy[i]) (void)hipEventDestroy(h->ev_copy[i]);
        if (h->ev_done[i]) (void)hipEventDestroy(h->ev_done[i]);
    }
    if (h->copy_st) (void)hipStreamDestroy(h->copy_st);
    delete h;
    return MDR_OK;
}

int mdr_index_reserve(mdr_index* h, int64_t n_rows) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(n_rows >= 0 && n_rows < 0xFFFFFFE0ll, "n_rows out of range");
    DeviceGuard g(h->device);
    return grow(h, n_rows, nullptr);
}

namespace {

// memcpy with `nt` threads (page-cache / mmap sources fault their pages in here, in parallel)
void parallel_copy(char* dst, const char* src, size_t bytes, int nt) {
    if (nt <= 1 || bytes < (8u << 20)) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t part = (bytes / (size_t)nt + 4095) & ~(size_t)4095;
    size_t done_to = part < bytes ? part : bytes;  // [0, part) is this thread's; a part whose thread cannot be started (resource limits) is copied here too
    for (int t = 1; t < nt; ++t) {
        const size_t lo = (size_t)t * part;
        if (lo >= bytes) break;
        const size_t len = bytes - lo < part ? bytes - lo : part;
        try {
            th.emplace_back([=] { memcpy(dst + lo, src + lo, len); });
        } catch (...) {  // std::system_error: no more threads -- nothing may throw across the C ABI
            memcpy(dst + lo, src + lo, len);
        }
    }
    memcpy(dst, src, done_to);
    for (auto& t : th) t.join();
}

// Host-sourced rows (pageable memory, np.load(mmap_mode="r") included) -> the shard, as a two-slot pipeline (round 5; VERDICT r4 item 8: the 256 MiB
// copy -> convert -> sync series took ~2 s of a 5.6 s process for a 15.4 GB index):
//   producer thread   rows -> pinned slot (parallel memcpy) -> hipMemcpyAsync to the slot's device half on the COPY stream -> "copied" event
//   calling thread    waits for "copied" on `st`, runs add_any (range fit + conversion kernels, as before), records "converted"
// so chunk c + 1 is read from the host and crosses PCIe while chunk c is converted; a slot is refilled when its "converted" event has completed.
// Returns an error code (the caller rolls the index back); `n` rows starting at logical row h->ntotal.
int upload_host_rows(mdr_index* h, const char* rows, long long n, int src_dtype, size_t row_src, hipStream_t st) {
    // Slots are sized by the call (a 100-row add used to allocate 2 x 96 MiB of device staging + 2 x 96 MiB of pinned host memory and start a thread):
    // chunk = min(n, 96 MiB / row) rows, and an add that fits ONE small chunk takes the plain road -- one hipMemcpyAsync out of the caller's (pageable) rows into a
    // staging buffer of exactly that size, the conversion, a sync; no pinned memory, no producer thread.
    const size_t chunk_bytes_target = 96ull << 20, plain_bytes_max = 16ull << 20;
    long long chunk_rows = (long long)(chunk_bytes_target / row_src) > 0 ? (long long)(chunk_bytes_target / row_src) : 1;
    if (n < chunk_rows) chunk_rows = n;
    const size_t chunk_bytes = (size_t)chunk_rows * row_src;
    const long long nchunks = (n + chunk_rows - 1) / chunk_rows;
    const bool plain = nchunks == 1 && chunk_bytes <= plain_bytes_max;
    const size_t stage_need = plain ? chunk_bytes : 2 * chunk_bytes;
    if (stage_need > h->stage_bytes) {
        if (h->stage) MDR_HIP_TRY(hipFree(h->stage));
        h->stage = nullptr;
        h->stage_bytes = 0;
        MDR_HIP_TRY(hipMalloc(&h->stage, stage_need));
        h->stage_bytes = stage_need;
    }
    if (plain) {
        MDR_HIP_TRY(hipMemcpyAsync(h->stage, rows, chunk_bytes, hipMemcpyHostToDevice, st));
        int rc_p = add_any(h, h->stage, src_dtype, n, h->ntotal, st);
        if (rc_p == MDR_OK && hipStreamSynchronize(st) != hipSuccess) rc_p = set_error(MDR_E_HIP, "stream sync failed in add()");
        return rc_p;
    }
    if (chunk_bytes > h->pin_bytes) {
        for (int i = 0; i < 2; ++i) {
            if (h->pin[i]) MDR_HIP_TRY(hipHostFree(h->pin[i]));
            h->pin[i] = nullptr;
        }
        h->pin_bytes = 0;
        for (int i = 0; i < 2; ++i) MDR_HIP_TRY(hipHostMalloc(&h->pin[i], chunk_bytes, hipHostMallocDefault));
        h->pin_bytes = chunk_bytes;
    }
    if (!h->copy_st) MDR_HIP_TRY(hipStreamCreateWithFlags(&h->copy_st, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!h->ev_copy[i]) MDR_HIP_TRY(hipEventCreateWithFlags(&h->ev_copy[i], hipEventDisableTiming));
        if (!h->ev_done[i]) MDR_HIP_TRY(hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
    }
    unsigned hc = std::thread::hardware_concurrency();
    const char* env = getenv("MDR_UPLOAD_THREADS");
    const int nt = env ? (atoi(env) > 0 ? atoi(env) : 1) : (hc >= 16 ? 8 : hc >= 4 ? (int)hc / 2 : 1);

    std::mutex mu;
    std::condition_variable cv;
    long long copied = 0, converted = 0;  // chunks whose H2D copy has been ISSUED / whose conversion has been ISSUED (its event recorded)
    bool abort_flag = false;
    int producer_rc = MDR_OK;
    const int device = h->device;
    auto produce = [&] {
        if (hipSetDevice(device) != hipSuccess) { std::lock_guard<std::mutex> g(mu); producer_rc = MDR_E_HIP; abort_flag = true; cv.notify_all(); return; }
        for (long long c = 0; c < nchunks; ++c) {
            const int slot = (int)(c & 1);
            if (c >= 2) {  // the slot's previous tenant (chunk c - 2) must have been converted: wait until that was issued, then until it completed
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return converted >= c - 1 || abort_flag; });
                if (abort_flag) return;
                lk.unlock();
                if (hipEventSynchronize(h->ev_done[slot]) != hipSuccess) { std::lock_guard<std::mutex> g(mu); producer_rc = MDR_E_HIP; abort_flag = true; cv.notify_all(); return; }
            }
            const long long r0 = c * chunk_rows, nr = n - r0 < chunk_rows ? n - r0 : chunk_rows;
            parallel_copy((char*)h->pin[slot], rows + (size_t)r0 * row_src, (size_t)nr * row_src, nt);
            bool ok = hipMemcpyAsync((char*)h->stage + (size_t)slot * chunk_bytes, h->pin[slot], (size_t)nr * row_src, hipMemcpyHostToDevice, h->copy_st) == hipSuccess;
            ok = ok && hipEventRecord(h->ev_copy[slot], h->copy_st) == hipSuccess;
            std::lock_guard<std::mutex> g(mu);
            if (!ok) { producer_rc = MDR_E_HIP; abort_flag = true; cv.notify_all(); return; }
            if (abort_flag) return;
            copied = c + 1;
            cv.notify_all();
        }
    };
    std::thread producer;
    try {
        producer = std::thread(produce);
    } catch (...) {  // no thread to be had: the same chunks, one after the other, on this thread (nothing may throw across the C ABI)
        int rc_s = MDR_OK;
        for (long long c = 0; c < nchunks && rc_s == MDR_OK; ++c) {
            const long long r0 = c * chunk_rows, nr = n - r0 < chunk_rows ? n - r0 : chunk_rows;
            memcpy(h->pin[0], rows + (size_t)r0 * row_src, (size_t)nr * row_src);
            if (hipMemcpyAsync(h->stage, h->pin[0], (size_t)nr * row_src, hipMemcpyHostToDevice, st) != hipSuccess) { rc_s = set_error(MDR_E_HIP, "hipMemcpyAsync(host rows) failed"); break; }
            rc_s = add_any(h, h->stage, src_dtype, nr, h->ntotal + r0, st);
            if (rc_s == MDR_OK && hipStreamSynchronize(st) != hipSuccess) rc_s = set_error(MDR_E_HIP, "stream sync failed in add()");
        }
        return rc_s;
    }
    int rc = MDR_OK;
    for (long long c = 0; c < nchunks && rc == MDR_OK; ++c) {
        const int slot = (int)(c & 1);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return copied > c || abort_flag; });
            if (abort_flag) { rc = producer_rc ? set_error(producer_rc, "host upload pipeline failed (copy side)") : MDR_E_HIP; break; }
        }
        const long long r0 = c * chunk_rows, nr = n - r0 < chunk_rows ? n - r0 : chunk_rows;
        if (hipStreamWaitEvent(st, h->ev_copy[slot], 0) != hipSuccess) { rc = set_error(MDR_E_HIP, "hipStreamWaitEvent failed in add()"); break; }
        rc = add_any(h, (char*)h->stage + (size_t)slot * chunk_bytes, src_dtype, nr, h->ntotal + r0, st);
        if (rc == MDR_OK && hipEventRecord(h->ev_done[slot], st) != hipSuccess) rc = set_error(MDR_E_HIP, "hipEventRecord failed in add()");
        std::lock_guard<std::mutex> g(mu);
        if (rc == MDR_OK) converted = c + 1;
        else abort_flag = true;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> g(mu);
        if (rc != MDR_OK) abort_flag = true;
        cv.notify_all();
    }
    producer.join();
    (void)hipStreamSynchronize(h->copy_st);  // nothing of the caller's buffer or the pinned slots is in flight after return
    if (rc == MDR_OK && hipStreamSynchronize(st) != hipSuccess) rc = set_error(MDR_E_HIP, "stream sync failed in add()");
    return rc;
}

}  // namespace

int mdr_index_add(mdr_index* h, const void* rows, int64_t n, int src_dtype, int rows_on_device, void* stream) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(n >= 0, "n < 0");
    MDR_REQUIRE(n == 0 || rows != nullptr, "rows is NULL");
    MDR_REQUIRE(src_dtype >= MDR_DT_F32 && src_dtype <= MDR_DT_F16, "unknown src_dtype %d", src_dtype);
    MDR_REQUIRE(h->ntotal + n < 0xFFFFFFE0ll, "index would exceed 2^32-32 rows per shard");
    if (n == 0) return MDR_OK;
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = grow(h, h->ntotal + n, st);
    if (rc) return rc;
    int before[16] = {0};  // [0..3] range / query / norm flags, [8..11] the int8 tier's row statistics and its query-split verdict: all restored when the rows are rejected
    const bool centre_was_set = h->centre_set;
    const int xexp_was = h->xexp;
    const bool xexp_was_set = h->xexp_set;
    MDR_HIP_TRY(hipMemcpyAsync(before, h->flags, sizeof(before), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    // ONE rollback for every failure from here on: a rejected add() -- in the first chunk or a later one, by the range check of
    // fit_exponent or by a conversion kernel's flag -- leaves the index exactly as it was: rows invisible (ntotal unchanged), norm / int8 statistics,
    // the int8 plane's centre and the exponent as before. (A rescale that an earlier chunk of the same call applied to the stored planes is exact
    // -- a power of two -- and stays; only the bookkeeping returns, so xexp is restored only when no rescale happened.)
    auto reject = [&](int code) {
        (void)hipMemcpyAsync(h->flags, before, sizeof(before), hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        h->centre_set = centre_was_set;
        h->cb = before[11] != 0;
        if (!xexp_was_set) { h->xexp = xexp_was; h->xexp_set = false; }  // the exponent was fitted to rejected rows: forget it
        return code;
    };
    const size_t row_src = (size_t)h->d * elem_size(src_dtype);
    if (rows_on_device) {
        rc = add_any(h, rows, src_dtype, n, h->ntotal, st);
        if (rc) return reject(rc);
    } else {
        rc = upload_host_rows(h, (const char*)rows, n, src_dtype, row_src, st);
        if (rc) return reject(rc);
    }
    int after[16] = {0};
    MDR_HIP_TRY(hipMemcpyAsync(after, h->flags, sizeof(after), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    if (after[0]) return reject(set_error(MDR_E_RANGE, "add(): a value is non-finite; rows were not added"));
    h->cb = after[11] != 0;  // (written once, with the centre: centre_finish_kernel)
    h->ntotal += n;
    return MDR_OK;
}

int mdr_upload_host(void* dst_dev, const void* src_host, size_t bytes, int device, void* stream) {
    MDR_REQUIRE(bytes == 0 || (dst_dev && src_host), "NULL pointer");
    if (bytes == 0) return MDR_OK;
    DeviceGuard g(device);
    if (!g.ok) return set_error(MDR_E_HIP, "hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)stream;
    const size_t chunk = 64ull << 20;
    if (bytes <= (4ull << 20)) {  // small: one plain copy
        MDR_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        return MDR_OK;
    }
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = MDR_OK;
    auto fail = [&](const char* what) { rc = set_error(MDR_E_HIP, "%s failed in mdr_upload_host", what); };
    for (int i = 0; i < 2 && rc == MDR_OK; ++i) {
        if (hipHostMalloc(&pin[i], chunk, hipHostMallocDefault) != hipSuccess) fail("hipHostMalloc");
        else if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) fail("hipEventCreate");
    }
    unsigned hc = std::thread::hardware_concurrency();
    const char* env = getenv("MDR_UPLOAD_THREADS");
    const int nt = env ? (atoi(env) > 0 ? atoi(env) : 1) : (hc >= 16 ? 8 : hc >= 4 ? (int)hc / 2 : 1);
    size_t off = 0;
    for (long long c = 0; rc == MDR_OK && off < bytes; ++c) {
        const int slot = (int)(c & 1);
        const size_t len = bytes - off < chunk ? bytes - off : chunk;
        if (c >= 2 && hipEventSynchronize(ev[slot]) != hipSuccess) { fail("hipEventSynchronize"); break; }  // the slot's previous copy has left the pinned buffer
        parallel_copy((char*)pin[slot], (const char*)src_host + off, len, nt);
        if (hipMemcpyAsync((char*)dst_dev + off, pin[slot], len, hipMemcpyHostToDevice, st) != hipSuccess) { fail("hipMemcpyAsync"); break; }
        if (hipEventRecord(ev[slot], st) != hipSuccess) { fail("hipEventRecord"); break; }
        off += len;
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == MDR_OK) fail("hipStreamSynchronize");
    for (int i = 0; i < 2; ++i) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (pin[i]) (void)hipHostFree(pin[i]);
    }
    return rc;
}

int64_t mdr_index_ntotal(const mdr_index* h) { return h ? h->ntotal : 0; }
int mdr_index_dim(const mdr_index* h) { return h ? h->d : 0; }
int64_t mdr_index_stream_bytes(const mdr_index* h) { return h ? (int64_t)((h->ntotal + 15) / 16 * 16) * (int64_t)h->d * (is_bf16(h) ? 2 : 4) : 0; }

int mdr_index_set_variant(mdr_index* h, int variant) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(variant >= 0 && variant <= (MDR_MIPS_GEMMK ? 5 : 4), "variant must be 0 (auto), 1 (generic), 2 (exact stream), 3 (screen + refine) or 4 (screen + refine "
                                                                      "without the int8 tier); 5 (GEMM-structured screen-k pass) exists in -DMDR_MIPS_GEMMK=1 builds only");
    h->variant = variant;
    return MDR_OK;
}

const char* mdr_index_last_kernel(const mdr_index* h) { return h ? h->last_kernel : "none"; }

int mdr_index_queries_per_pass(const mdr_index* h, int nq, int k) {
    if (!h || k < 1 || k > kKMax || nq < 1) return 0;
    const int path = make_plan(h, nq, k).path;
    if (path == PATH_GENERIC) return kGenericQ;
    return path == PATH_SCREEN && wide_pass(nq) ? kWideQ : kStreamQ;
}

size_t mdr_index_search_workspace_bytes(const mdr_index* h, int nq, int k) {
    if (!h || nq < 0 || k < 1 || k > kKMax) return 0;
    return make_plan(h, nq, k).total;
}

int mdr_index_search(mdr_index* h, const float* q_dev, int nq, int k, float* D_dev, int64_t* I_dev, int64_t id_offset, void* workspace_dev,
                     size_t workspace_bytes, void* stream) {
    MDR_REQUIRE(h != nullptr, "index handle is NULL");
    MDR_REQUIRE(nq >= 0, "nq < 0");
    MDR_REQUIRE(k >= 1 && k <= kKMax, "k=%d out of range [1, %d]", k, kKMax);
    if (nq == 0) return MDR_OK;
    MDR_REQUIRE(q_dev && D_dev && I_dev, "NULL query/result pointer");
    if (h->variant == PATH_STREAM && !stream_kernel_supports(h, k))
        return set_error(MDR_E_INVALID, "stream kernel forced but unsupported for d=%d k=%d storage=%d (needs F32X2H, d=768, k<=128)", h->d, k, h->storage);
    if (h->variant == PATH_SCREEN && !screen_kernel_supports(h, k))
        return set_error(MDR_E_INVALID, "screen kernel forced but unsupported for d=%d k=%d (needs d=768, k<=128)", h->d, k);
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    long long* I_ll = (long long*)I_dev;
    if (h->ntotal == 0) {
        long long n = (long long)nq * k;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, D_dev, I_ll, n);
        MDR_HIP_TRY(hipGetLastError());
        h->last_kernel = "fill_empty_kernel";
        return MDR_OK;
    }
    SearchPlan p = make_plan(h, nq, k);
    if (!workspace_dev || workspace_bytes < p.total)
        return set_error(MDR_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p.total, workspace_bytes);
    char* ws = (char*)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    const bool bf = is_bf16(h);
    int rc;
    if (p.path == PATH_GENERIC) {
        rc = bf ? run_generic<true>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, nullptr, st)
                : run_generic<false>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, nullptr, st);
        h->last_kernel = "mips_generic_kernel";
        return rc;
    }

    constexpr int NKB = 24;
    const size_t rb_bytes = (size_t)NKB * 2 * kFragBytes;  // one exact-kernel stage == one screen-kernel stage == 48 KiB
    const int n_rb = (int)((h->ntotal + 15) / 16);
    const int ngroups = (nq + kStreamQ - 1) / kStreamQ;
    const size_t qgroup_bytes = (size_t)kStreamQ * h->d * 2;
    u64* best = (u64*)(ws + p.off_best);
    char* qhi = ws + p.off_qhi;
    char* qlo = ws + p.off_qlo;
    float* qscale = (float*)(ws + p.off_qscale);
    {
        // |q.x - qh.xh| <= c |q| max|x|: fp16 rounding of both operands (2^-10) or bf16 rounding of q only (2^-9; the
        // stored rows ARE the bf16 values), plus fp32 accumulation slack
        const float c = bf ? 2.2e-3f : 1.2e-3f;
        const int nq_pad = (nq + kWideQ - 1) / kWideQ * kWideQ;
        float* bound = (float*)(ws + p.off_bound);
        MDR_HIP_TRY(hipMemsetAsync(h->flags + 1, 0, sizeof(int), st));  // "a query of THIS call was non-finite" (telemetry)
        if (bf)
            hipLaunchKernelGGL((prep_queries_kernel<true>), dim3((nq_pad + 3) / 4), dim3(256), 0, st, q_dev, nq, nq_pad, h->d, h->flags, c, qhi, qlo, bound, qscale, 1.0f);
        else if (p.i8 && k == 1) {
            const int nq_pad8 = wide_pass(nq) ? nq_pad : kStreamQ;  // what run_screen8w / run_screen8 read: whole groups of 256 / one group of 128 queries
            const int nb16 = (nq_pad + 3) / 4, nb8 = (nq_pad8 + 3) / 4;
            hipLaunchKernelGGL(prep_queries_both_kernel, dim3(nb16 + nb8), dim3(256), 0, st, nb16, q_dev, nq, nq_pad, nq_pad8, h->d, h->flags, c, qhi, qlo, bound, qscale,
                               row_unscale(h), ws + p.off_q8, (f32x4*)(ws + p.off_qab), (const float*)h->centre, (int)use_cb(h), (float*)(ws + p.off_qlam));
        } else
            hipLaunchKernelGGL((prep_queries_kernel<false>), dim3((nq_pad + 3) / 4), dim3(256), 0, st, q_dev, nq, nq_pad, h->d, h->flags, c, qhi, qlo, bound, qscale,
                               row_unscale(h));
        MDR_HIP_TRY(hipGetLastError());
    }

    if (k == 1) {
        if (p.path == PATH_SCREEN) MDR_HIP_TRY(hipMemsetAsync(ws + p.off_zero, 0, p.zero_bytes, st));  // best + every tier's control words, one node
        else MDR_HIP_TRY(hipMemsetAsync(best, 0, (size_t)nq * 8, st));
        const int* run_if = nullptr;
        if (p.path == PATH_SCREEN) {
            if (wide_pass(nq) && p.i8) {  // int8 tier, 256 queries per pass; the fp16 wide screen only behind an overflow
                rc = use_cb(h) ? run_screen8w<true>(h, p, ws, q_dev, nq, best, st) : run_screen8w<false>(h, p, ws, q_dev, nq, best, st);
                if (!rc) rc = run_screen32<false>(h, p, ws, q_dev, nq, qhi, best, st, (const int*)(ws + p.off_ctl8));
                h->last_kernel = "mips_screen8w_kernel<12,1>";
            } else if (wide_pass(nq)) {  // more than 128 queries: 256 per corpus pass on the 32-queries-per-wave kernel
                rc = bf ? run_screen32<true>(h, p, ws, q_dev, nq, qhi, best, st) : run_screen32<false>(h, p, ws, q_dev, nq, qhi, best, st);
                h->last_kernel = bf ? "mips_screen32_kernel<24,1,bf16>" : "mips_screen32_kernel<24,1>";
            } else if (p.i8) {  // int8 tier first; the fp16 screen only if one of its lists overflowed, the exact pass only if that one's did
                rc = use_cb(h) ? run_screen8<true>(h, p, ws, q_dev, nq, best, st) : run_screen8<false>(h, p, ws, q_dev, nq, best, st);
                if (!rc) rc = run_screen<false>(h, p, ws, q_dev, nq, qhi, best, st, (const int*)(ws + p.off_ctl8));
                h->last_kernel = "mips_screen8_kernel<12,1>";
            } else {
                rc = bf ? run_screen<true>(h, p, ws, q_dev, nq, qhi, best, st) : run_screen<false>(h, p, ws, q_dev, nq, qhi, best, st);
                h->last_kernel = bf ? "mips_screen_kernel<24,1,bf16>" : "mips_screen_kernel<24,1>";
            }
            if (rc) return rc;
            run_if = (const int*)(ws + p.off_sctl);  // exact pass below: only if a candidate list overflowed
        } else {
            h->last_kernel = "mips_stream_kernel<24,0>";
        }
        if (!bf) {
            rc = ensure_dynamic_lds((const void*)mips_stream_kernel<NKB, 0>, (int)(3 * rb_bytes));
            if (rc) return rc;
            const int Gx = (int)(n_rb < h->num_cus ? n_rb : h->num_cus);
            for (int gi = 0; gi < ngroups; ++gi) {
                const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
                hipLaunchKernelGGL((mips_stream_kernel<NKB, 0>), dim3(Gx), dim3(512), 3 * rb_bytes, st, (const char*)h->hi, (const char*)h->lo,
                                   (long long)h->ntotal, n_rb, (const char*)(qhi + gi * qgroup_bytes), (const char*)(qlo + gi * qgroup_bytes), nqg,
                                   best + (size_t)gi * kStreamQ, (u64*)nullptr, (int*)nullptr, (u64*)nullptr, 1, run_if,
                                   (const float*)(qscale + (size_t)gi * kStreamQ));
                MDR_HIP_TRY(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(finalize_top1_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, (const u64*)best, nq, D_dev, I_ll, (long long)id_offset);
        MDR_HIP_TRY(hipGetLastError());
        if (bf)  // bf16 storage has no exact MFMA pass: the overflow fallback is the generic kernel, overwriting D/I
            return run_generic<true>(h, p, ws, q_dev, nq, 1, D_dev, I_ll, id_offset, run_if, st);
        return MDR_OK;
    }

    // 2 <= k <= 256 (screen path; the exact stream kernel behind / instead of it serves k <= 128)
    const int* run_if = nullptr;
    if (p.path == PATH_SCREEN) {
        if (wide_pass(nq)) {
            rc = bf ? run_screenk32<true>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st)
                    : run_screenk32<false>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st);
            if (gemmk_on(h) && h->ntotal >= 256ll * p.G) h->last_kernel = bf ? "mips_gemmk_kernel<bf16>" : "mips_gemmk_kernel";
            else h->last_kernel = bf ? "mips_screenk32_kernel<24,bf16>" : "mips_screenk32_kernel<24>";
        } else {
            rc = bf ? run_screenk<true>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st)
                    : run_screenk<false>(h, p, ws, q_dev, nq, k, qhi, D_dev, I_ll, id_offset, st);
            h->last_kernel = bf ? "mips_screenk_kernel<24,bf16>" : "mips_screenk_kernel<24>";
        }
        if (rc) return rc;
        run_if = (const int*)(ws + p.off_sctl);  // exact pass below: only if a list or the band overflowed
        if (bf) return run_generic<true>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, run_if, st);
        if (!stream_kernel_supports(h, k)) return run_generic<false>(h, p, ws, q_dev, nq, k, D_dev, I_ll, id_offset, run_if, st);
    } else {
        h->last_kernel = "mips_stream_kernel<24,1>";
    }
    // F32X2H: exact stream kernel with candidate lists (unconditional on the stream path)
    u64* cand = (u64*)(ws + p.off_cand);
    int* cnt = (int*)(ws + p.off_cnt);
    u64* kth = (u64*)(ws + p.off_kth);
    const size_t lds_bytes = 3 * rb_bytes + kStreamQ * sizeof(int);
    rc = ensure_dynamic_lds((const void*)mips_stream_kernel<NKB, 1>, (int)lds_bytes);
    if (rc) return rc;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int nqg = nq - gi * kStreamQ < kStreamQ ? nq - gi * kStreamQ : kStreamQ;
        MDR_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)p.Gx * kStreamQ * 4, st));
        MDR_HIP_TRY(hipMemsetAsync(kth, 0, (size_t)p.Gx * kStreamQ * 8, st));
        hipLaunchKernelGGL((mips_stream_kernel<NKB, 1>), dim3(p.Gx), dim3(512), lds_bytes, st, (const char*)h->hi, (const char*)h->lo, (long long)h->ntotal,
                           n_rb, (const char*)(qhi + gi * qgroup_bytes), (const char*)(qlo + gi * qgroup_bytes), nqg, (u64*)nullptr, cand, cnt, kth, k,
                           run_if, (const float*)(qscale + (size_t)gi * kStreamQ));
        MDR_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(merge_lists_kernel, dim3(nqg), dim3(256), 0, st, (const u64*)cand, (const int*)cnt, (const u64*)kth, p.Gx, kStreamQ, kStreamCap, k,
                           D_dev + (size_t)gi * kStreamQ * k, I_ll + (size_t)gi * kStreamQ * k, (long long)id_offset, run_if,
                           (const float*)(qscale + (size_t)gi * kStreamQ));
        MDR_HIP_TRY(hipGetLastError());
    }
    return MDR_OK;
}

#if MDR_I8_ABL == 9  // measurement builds only (include/mdr_hip_measure.h)
int mdr_test_i8_stamps(unsigned long long* out8_host, int reset) {
    MDR_REQUIRE(out8_host, "NULL pointer");
    MDR_HIP_TRY(hipDeviceSynchronize());
    MDR_HIP_TRY(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_i8_stamp), 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        MDR_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamp), z, sizeof(z)));
    }
    return MDR_OK;
}
#endif

int mdr_index_search_telemetry(const mdr_index* h, int nq, int k, const void* workspace_dev, int64_t* out4_host, void* stream) {
    MDR_REQUIRE(h && workspace_dev && out4_host, "NULL argument");
    MDR_REQUIRE(nq >= 1 && k >= 1 && k <= kKMax, "bad shape");
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    const SearchPlan p = make_plan(h, nq, k);
    const char* ws = (const char*)(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    out4_host[0] = out4_host[1] = out4_host[2] = 0;
    out4_host[3] = p.path;
    int flags[4] = {0, 0, 0, 0};
    MDR_HIP_TRY(hipMemcpyAsync(flags, h->flags, sizeof(flags), hipMemcpyDeviceToHost, st));
    MDR_HIP_TRY(hipStreamSynchronize(st));
    out4_host[2] = flags[1];
    if (p.path != PATH_SCREEN) return MDR_OK;
    int overflow = 0;
    MDR_HIP_TRY(hipMemcpyAsync(&overflow, ws + p.off_sctl, sizeof(int), hipMemcpyDeviceToHost, st));
    long long total = 0;
    if (k == 1) {  // per-wave list lengths of the last query group
        const size_t n_cnt = (size_t)(p.i8 ? (wide_pass(nq) ? p.G8w : p.G8) : p.G) * 8;
        int* cnt = new (std::nothrow) int[n_cnt];
        MDR_REQUIRE(cnt != nullptr, "out of host memory");
        hipError_t e = hipMemcpyAsync(cnt, ws + p.off_sctl + 256, n_cnt * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        for (size_t i = 0; i < n_cnt; ++i) total += cnt[i];
        delete[] cnt;
        if (e != hipSuccess) return set_error(MDR_E_HIP, "telemetry copy failed: %s", hipGetErrorString(e));
    } else {  // merge_screenk_kernel adds every query's union size to sctl[1]
        int t = 0;
        MDR_HIP_TRY(hipMemcpyAsync(&t, ws + p.off_sctl + sizeof(int), sizeof(int), hipMemcpyDeviceToHost, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        total = t;
    }
    out4_host[0] = overflow;
    out4_host[1] = total;
    if (p.i8) {  // bit 9: the int8 tier ran in front; bit 8: one of its lists overflowed (the fp16 screen ran behind it)
        int o8 = 0;
        MDR_HIP_TRY(hipMemcpyAsync(&o8, ws + p.off_ctl8, sizeof(int), hipMemcpyDeviceToHost, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        int kept = 0;
        MDR_HIP_TRY(hipMemcpyAsync(&kept, ws + p.off_ctl8 + sizeof(int), sizeof(int), hipMemcpyDeviceToHost, st));
        MDR_HIP_TRY(hipStreamSynchronize(st));
        out4_host[3] |= 512 | (o8 ? 256 : 0) | (use_cb(h) ? 1024 : 0) | ((int64_t)kept << 16);  // bit 10: the query split is on; bits 16..: candidates of the int8 tier that were really re-scored
    }
    return MDR_OK;
}

int mdr_topk_merge(const float* D_parts_dev, const int64_t* I_parts_dev, int nparts, int nq, int k, float* D_dev, int64_t* I_dev, void* stream) {
    MDR_REQUIRE(nparts >= 1 && nq >= 0 && k >= 1 && k <= kKMax, "bad merge shape nparts=%d nq=%d k=%d", nparts, nq, k);
    if (nq == 0) return MDR_OK;
    MDR_REQUIRE(D_parts_dev && I_parts_dev && D_dev && I_dev, "NULL pointer");
    hipLaunchKernelGGL(merge_parts_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, D_parts_dev, (const long long*)I_parts_dev, nparts, nq, k, D_dev,
                       (long long*)I_dev);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

size_t mdr_topk_packed_ids_offset(int nq, int k) {
    if (nq < 0 || k < 1) return 0;
    return align_up((size_t)nq * (size_t)k * 4, 8);
}

size_t mdr_topk_packed_bytes(int nq, int k) {
    if (nq < 0 || k < 1) return 0;
    return align_up(mdr_topk_packed_ids_offset(nq, k) + (size_t)nq * (size_t)k * 8, 256);
}

int mdr_topk_merge_packed(const void* packed_parts_dev, int nparts, int nq, int k, float* D_dev, int64_t* I_dev, void* stream) {
    MDR_REQUIRE(nparts >= 1 && nq >= 0 && k >= 1 && k <= kKMax, "bad merge shape nparts=%d nq=%d k=%d", nparts, nq, k);
    if (nq == 0) return MDR_OK;
    MDR_REQUIRE(packed_parts_dev && D_dev && I_dev, "NULL pointer");
    MDR_REQUIRE(((uintptr_t)packed_parts_dev & 7) == 0, "packed blocks must be 8-byte aligned");
    hipLaunchKernelGGL(merge_packed_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, (const char*)packed_parts_dev, (long long)mdr_topk_packed_bytes(nq, k),
                       (long long)mdr_topk_packed_ids_offset(nq, k), nparts, nq, k, D_dev, (long long*)I_dev);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

}  // extern "C"

/*
 * include/mdr_hip.h -- C ABI of libmdrhip.so, the MI355X (gfx950) implementation of the
 * iterative-retrieval hot path of facebookresearch/multihop_dense_retrieval.
 *
 * The reference has no FFI of its own (it is pure Python). Each entry point below
 * replaces the third-party native call the reference makes at the cited line, with plain C types
 * only (no torch / numpy / faiss types), so that any host language can bind it. INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = error class (MDR_E_*); the message of the last
 *     error on the calling thread is mdr_last_error(). Nothing throws, nothing calls exit().
 *   - pointers named *_dev are DEVICE pointers on the handle's device; *_host are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). All work is enqueued
 *     asynchronously on it; no entry point synchronises the device except mdr_index_add /
 *     mdr_index_reserve (index construction, not the search loop): add() waits on `stream` for host AND
 *     device sources, because it validates the rows (range / non-finite) before it makes them visible
 *     (ntotal) and must have consumed a host buffer before it returns. It must therefore not be called
 *     while `stream` is being captured into a hipGraph; search / merge / encoder / assemble may be.
 *   - the caller owns every buffer it passes (queries, results, workspace); a handle owns only its
 *     own storage (corpus shard, encoder weight copies).
 *   - one handle per (process, device); calls on one handle are not re-entrant.
 */
#ifndef MDR_HIP_H
#define MDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDR_OK 0
#define MDR_E_INVALID (-1)  /* bad argument (k < 1, k > MDR_KMAX, nq < 0, d unsupported, NULL ...) */
#define MDR_E_HIP (-2)      /* a HIP runtime call failed; mdr_last_error() carries hipGetErrorString */
#define MDR_E_RANGE (-3)    /* a row value is non-finite (add() takes any finite fp32 like IndexFlatIP.add: F32X2H rows are stored times a
                             * power of two fitted to the data) */
#define MDR_E_WORKSPACE (-4) /* workspace too small: call the matching *_workspace_bytes first */
#define MDR_E_STATE (-5)    /* handle in the wrong state (e.g. search on an empty encoder) */

#define MDR_KMAX 1024       /* largest k accepted by mdr_index_search */

/* element types of caller-side arrays */
#define MDR_DT_F32 0
#define MDR_DT_BF16 1
#define MDR_DT_F16 2

/* index storage formats */
#define MDR_STORE_F32X2H 0  /* fp32-accurate: each element kept as an fp16 (hi, lo) pair = 4 bytes; d = 768 indexes also carry an int8
                             * screening copy (1 byte per element + 4 per row, +25 % device memory) that k = 1 searches stream first --
                             * results do not depend on it (environment MDR_MIPS_I8=0 at creation leaves it out) */
#define MDR_STORE_BF16 1    /* rows rounded to bf16 (RNE), 2 bytes per element; scores exact w.r.t. the rounded rows */
#define MDR_STORE_F32X2H_COMPACT 2  /* MDR_STORE_F32X2H without the int8 screening copy: exactly faiss.IndexFlatIP's 4 bytes per element of device
                             * memory; same results, k = 1 searches start at the fp16 screen (about 1.5 x the search time at d = 768) */

const char* mdr_last_error(void);
const char* mdr_version(void);

/* ------------------------------------------------------------------------------------------------
 * Flat inner-product index  ==  faiss.IndexFlatIP as used by the reference:
 *   index = faiss.IndexFlatIP(d)          /root/reference/scripts/eval/eval_mhop_retrieval.py:121
 *   index.add(xb)                         /root/reference/scripts/eval/eval_mhop_retrieval.py:122
 *   index_cpu_to_gpu(res, 6, index)       /root/reference/scripts/eval/eval_mhop_retrieval.py:123-125
 *   D, I = index.search(q, beam)          /root/reference/scripts/eval/eval_mhop_retrieval.py:155,179
 * ---------------------------------------------------------------------------------------------- */
typedef struct mdr_index mdr_index;

/* d: vector dimension (multiple of 32, <= 1024). storage: MDR_STORE_*. device: HIP device ordinal. */
int mdr_index_create(int d, int storage, int device, mdr_index** out);
int mdr_index_free(mdr_index* h);

/* Pre-size the shard (rows). Optional: add() grows geometrically, but reserving avoids the copy. */
int mdr_index_reserve(mdr_index* h, int64_t n_rows);

/* Append n rows of `d` elements (C-contiguous, element type src_dtype) from a host or device buffer.
 * Rows receive consecutive ids starting at mdr_index_ntotal(). (IndexFlatIP.add)
 * A rejected call (MDR_E_RANGE: a non-finite value, in any chunk of a host-sourced add) leaves the index exactly as it was.
 * Dynamic range of MDR_STORE_F32X2H: rows are stored times ONE power of two per index, fitted to the first add() and grown (exactly) when a
 * later add() holds values >= 2^15 times larger; an element keeps fp32 accuracy while it lies within ~2^-14 of the index-wide maximum
 * |x| (its fp16 hi/lo pair is then normal) and degrades gracefully below that (lo, then hi, go subnormal: relative accuracy 1e-3..1e-4
 * at 2^-24 of the maximum, zero below ~2^-34). Embedding matrices (unit-scale LayerNorm outputs, the reference's case) sit inside the
 * accurate range by 10 orders of magnitude; an index mixing rows of wildly different scales is better served by MDR_STORE_BF16 or by
 * scaling the rows. */
int mdr_index_add(mdr_index* h, const void* rows, int64_t n, int src_dtype, int rows_on_device, void* stream);

int64_t mdr_index_ntotal(const mdr_index* h);
int mdr_index_dim(const mdr_index* h);
/* bytes of HBM one search call streams for the corpus (= ntotal_padded * d * bytes/elem) */
int64_t mdr_index_stream_bytes(const mdr_index* h);

/* Queries one corpus pass of mdr_index_search(nq, k) serves (the call streams the shard ceil(nq / this) times): the
 * accounting bench.py's roofline uses. 128 for up to 128 queries, 256 beyond (32 queries per wave). */
int mdr_index_queries_per_pass(const mdr_index* h, int nq, int k);

/* Workspace one search call needs (device memory, 256-byte aligned, contents don't persist). */
size_t mdr_index_search_workspace_bytes(const mdr_index* h, int nq, int k);

/* Top-k rows by inner product, best first (IndexFlatIP.search):
 *   q_dev  f32 [nq, d] C-contiguous     D_dev f32 [nq, k] descending     I_dev i64 [nq, k]
 * Ties are ordered by ascending row id (FAISS keeps the first-seen = lowest id among equal scores).
 * When the index holds fewer than k rows the tail is D = -FLT_MAX, I = -1 (FAISS behaviour).
 * id_offset is added to every returned id (global id of this shard's row 0). */
int mdr_index_search(mdr_index* h, const float* q_dev, int nq, int k, float* D_dev, int64_t* I_dev,
                     int64_t id_offset, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Test hook (synchronises `stream`): what the LAST search call of this shape left in its workspace.
 *   out4_host[0] = 1 when the hi-plane screen gave up (a candidate list or the refinement band overflowed) and the exact
 *                  pass re-did the search, else 0;  [1] = candidates the screen pass handed to the exact re-scoring (k == 1: of the
 *                  last group of 128 queries; k > 1: summed over all queries of the call);  [2] = 1 when a query held a non-finite value;  [3] = path (1 generic,
 *                  2 exact stream, 3 screen) in its low byte, | 512 when the int8 screening tier ran in front of the fp16 screen, | 256 when
 *                  one of that tier's candidate lists overflowed (the fp16 screen then ran behind it). Results never depend on it: it lets the parity tests assert that adversarial
 *                  inputs (all-ties corpora, large common mean) really took the fallback and benign ones did not. */
int mdr_index_search_telemetry(const mdr_index* h, int nq, int k, const void* workspace_dev, int64_t* out4_host, void* stream);

/* Test hook: force a kernel variant (0 = auto, 1 = generic fp32 reference kernel, 2 = exact MFMA stream kernel,
 * 3 = screen + exact refinement, 4 = the same without the int8 screening tier in front of the fp16 hi-plane screen, 5 = the screen path with the
 * GEMM-structured screen-k pass (csrc/mdr_mips_gemmk.inl; slower than the default 32-queries-per-wave kernel, kept with its parity test) for groups of 256
 * queries, 2 <= k <= 256). */
int mdr_index_set_variant(mdr_index* h, int variant);
/* Name of the kernel the last search dispatched to (for rocprof matching); static storage. */
const char* mdr_index_last_kernel(const mdr_index* h);

/* ------------------------------------------------------------------------------------------------
 * k-way merge of per-shard result lists (what every rank runs after the RCCL all-gather that
 * BASELINE.json's north_star adds in front of the hop-2 re-query; the reference itself is
 * single-index, eval_mhop_retrieval.py:155):
 *   D_parts f32 [nparts, nq, k], I_parts i64 [nparts, nq, k]  ->  D f32 [nq, k], I i64 [nq, k]
 * Same order rule as search: score descending, then id ascending; entries with I < 0 are padding.
 * ---------------------------------------------------------------------------------------------- */
int mdr_topk_merge(const float* D_parts_dev, const int64_t* I_parts_dev, int nparts, int nq, int k,
                   float* D_dev, int64_t* I_dev, void* stream);

/* The same merge over PACKED per-rank blocks, so that the exchange needs no glue kernels: a rank's search writes its (D, I) straight
 * into one block of mdr_topk_packed_bytes(nq, k) bytes -- scores f32 [nq, k] at offset 0, ids i64 [nq, k] at
 * mdr_topk_packed_ids_offset(nq, k) -- ONE all-gather of that block over the ranks (RCCL over xGMI) delivers
 * packed_parts_dev = nparts consecutive blocks, and this call merges them (8-byte aligned; order rule as above). */
size_t mdr_topk_packed_bytes(int nq, int k);
size_t mdr_topk_packed_ids_offset(int nq, int k);
int mdr_topk_merge_packed(const void* packed_parts_dev, int nparts, int nq, int k, float* D_dev, int64_t* I_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RoBERTa-base encoder forward + CLS projection  ==
 *   RobertaRetriever.encode_q / encode_seq   /root/reference/mdr/retrieval/models/mhop_retriever.py:23-26,40-41
 *   RobertaCtxEncoder.forward                /root/reference/mdr/retrieval/models/retriever.py:186-190
 * i.e. project(encoder(input_ids, mask)[0][:, 0, :]) with HF RobertaModel semantics, fp16 tensor-core
 * GEMMs with fp32 accumulation and fp32 LayerNorm / softmax / GELU (the apex-O1 numerics the
 * reference runs under, eval_mhop_retrieval.py:88-89).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mdr_encoder mdr_encoder;

typedef struct mdr_encoder_config {
    int vocab, hidden, layers, heads, ffn, max_pos, pad_id;
    float ln_eps;
    /* 1: LayerNorm outputs are kept in fp32 for the residual adds, only the copy that feeds the next Linear is rounded to
     *    fp16 -- the apex-O1 regime the reference runs under (eval_mhop_retrieval.py:88-89: LayerNorm is an fp32 op, the
     *    residual add promotes to fp32). 0: the residual stream is the fp16 copy (less HBM traffic, one more rounding per
     *    LayerNorm). 2: fp32 residual stream as in 1, and the out-projection / FFN2 outputs are rounded to fp16 BEFORE the residual
     *    add -- literally what apex O1 computes there (F.linear returns fp16; `hidden + input_tensor` promotes to fp32) and half the
     *    bytes in and out of those GEMM epilogues. Results differ inside fp16-operand noise; DESIGN.md §4 has the measured cost and
     *    error of each. */
    int residual_fp32;
} mdr_encoder_config;

/* One fp32 tensor of the q_encoder.pt schema, Linear weights [out, in]. */
typedef struct mdr_tensor {
    const char* name;   /* state-dict key without any "module." prefix                      */
    const float* data;  /* host or device pointer (see weights_on_device), C-contiguous fp32 */
    int64_t numel;
} mdr_tensor;

/* Copies (and converts to fp16 where the GEMMs want it) every tensor it needs; unknown names are
 * ignored (load_saved(exact=False), mdr/retrieval/utils/utils.py:19), a missing one is MDR_E_INVALID
 * naming the key (strict load_state_dict raises, utils.py:21). encoder.pooler.* is never required. */
int mdr_encoder_create(const mdr_encoder_config* cfg, const mdr_tensor* tensors, int n_tensors,
                       int weights_on_device, int device, void* stream, mdr_encoder** out);
int mdr_encoder_free(mdr_encoder* h);

size_t mdr_encoder_workspace_bytes(const mdr_encoder* h, int batch, int seq_len);

/* ids_dev / mask_dev: int64 [batch, seq_len] (right-padded; mask 1 = token, 0 = pad);
 * out_dev: f32 [batch, hidden]. Padded positions are skipped, which is result-identical for the
 * CLS embedding. */
int mdr_encoder_forward(mdr_encoder* h, const int64_t* ids_dev, const int64_t* mask_dev, int batch, int seq_len,
                        float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Optional: the fraction of non-pad positions (tokens / (batch * seq_len)) the caller expects in the next forwards, 0 = unknown.
 * Only steers tile-shape heuristics (the packed token count itself is computed on the device); results do not depend on it. */
int mdr_encoder_set_fill_hint(mdr_encoder* h, float fill);

/* ------------------------------------------------------------------------------------------------
 * Device-side construction of the hop-2 encoder inputs ==
 *   doc = id2doc[str(doc_id)]["text"]; empty -> title and D[b][j] = -inf     /root/reference/scripts/eval/eval_mhop_retrieval.py:158-166
 *   tokenizer.batch_encode_plus(pairs, max_length=max_q_sp_len, pad_to_max_length=True)                            :168
 * from a corpus tokenised once ("token arena"): arena_tokens i32 [sum len], arena_offsets i64 [n_docs+1], arena_empty u8
 * [n_docs] (1 = the passage text was empty, its arena entry is the title; may be NULL). Row b*beam+j of the output is
 *   <s> Q_b </s></s> D_{doc_ids[b*beam+j]} </s> <pad>...   with HF `longest_first` truncation to out_len,
 * Q_b = q_ids[b, 1 : len_b-1] (the hop-1 row without its <s> / </s>). hop1_scores (f32 [batch*beam], may be NULL) gets
 * -inf where arena_empty is set. Everything stays on the device: no sync, no host tokenizer between the hops.
 * ---------------------------------------------------------------------------------------------- */
int mdr_assemble_hop2(const int64_t* q_ids_dev, const int64_t* q_mask_dev, int batch, int q_len, const int64_t* doc_ids_dev, int beam,
                      const int32_t* arena_tokens_dev, const int64_t* arena_offsets_dev, const uint8_t* arena_empty_dev, int64_t n_docs,
                      float* hop1_scores_dev, int out_len, int bos_id, int eos_id, int pad_id, int64_t* out_ids_dev, int64_t* out_mask_dev,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Host array -> device buffer, pipelined (on-disk formats: the memory-mapped token arena and any other large host array) ==
 *   batch_q_encodes = move_to_cuda(dict(batch_q_encodes))                    /root/reference/mdr/retrieval/utils/utils.py:24-41 (`.cuda()` of a host tensor)
 * for arrays of gigabytes in PAGEABLE memory (np.load(mmap_mode="r") included): two pinned staging buffers are filled by parallel memcpy while the
 * previous chunk crosses PCIe (the same pipeline mdr_index_add runs for host rows). Synchronises `stream` before it returns: the
 * source may be unmapped and the destination read afterwards.
 * ---------------------------------------------------------------------------------------------- */
int mdr_upload_host(void* dst_dev, const void* src_host, size_t bytes, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Test hook: one encoder GEMM in isolation, out[M, N] = epilogue(A[M, K] . W[N, K]^T + bias[N]) -- the torch.nn.Linear
 * inside HF RobertaModel's layers (the reference reaches them through model.encode_q, mhop_retriever.py:23-26).
 * A, W: f16 device, K-contiguous; bias f32. epilogue: 0 = bias -> f16 out, 1 = bias + erf-GELU -> f16 out,
 * 3 = bias -> f32 out. kernel: 0 = the shape heuristic the encoder uses, 1 = 64x64 tiles, 2 = 128x128, 4 = persistent
 * 256x128, 6 = persistent 256x256 (N % 256 == 0). m_dev (may be NULL) is a device int holding the
 * number of valid rows (<= M), as the packed token count is only known on the device.
 * ---------------------------------------------------------------------------------------------- */
int mdr_test_gemm_f16(const void* A_dev, const void* W_dev, const float* bias_dev, int M, const int* m_dev, int N, int K,
                      void* out_dev, int epilogue, int kernel, int device, void* stream);
/* Measurement hooks that exist only in VARIANT builds of these sources (never in the product library) are declared in
 * include/mdr_hip_measure.h. No environment variable changes what any entry point above computes: MDR_GEMM_CFG, MDR_MIPS_WIDE,
 * MDR_MIPS_I8, MDR_MIPS_I8_CB (the int8 tier's query split forced on / off) and MDR_MIPS_EVEN_GROUPS (how the passes of a > 256-query
 * call share the queries) only choose between kernels / schedules that return the same bits / the same exact results
 * (tests/test_capi_symbols.py keeps the list of getenv() names in csrc/ closed). */

#ifdef __cplusplus
}
#endif
#endif /* MDR_HIP_H */

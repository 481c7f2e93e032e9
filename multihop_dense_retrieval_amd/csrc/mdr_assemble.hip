// csrc/mdr_assemble.hip -- device-side construction of the hop-2 encoder inputs from a pre-tokenised corpus.
//
// Replaces the host round trip between the two hops of /root/reference/scripts/eval/eval_mhop_retrieval.py:158-169
//   doc = id2doc[str(doc_id)]["text"]   (empty text -> title and D[b][j] = -inf, :162-165)
//   tokenizer.batch_encode_plus([(question, doc), ...], max_length=max_q_sp_len, pad_to_max_length=True)
// i.e. RoBERTa pair encoding  <s> Q </s></s> D </s>  with `longest_first` truncation (transformers 2.11
// truncate_sequences: remove one token at a time from the longer of the two, the pair when they tie) and right
// padding with <pad> to max_length. The corpus side is tokenised ONCE (token arena: int32 tokens + int64 offsets,
// plus a flag for passages whose text was empty and whose arena entry is therefore the title).
#include "mdr_common.h"

namespace mdr {
namespace {

// tokens to keep of (a = question, b = passage) after removing r, HF longest_first order
__host__ __device__ inline void longest_first(int& a, int& b, int r) {
    if (r <= 0) return;
    if (a > b) {
        int d = r < a - b ? r : a - b;
        a -= d;
        r -= d;
        b -= (r + 1) / 2;  // a == b now: the pair loses the next token first
        a -= r / 2;
    } else {
        int d = r < b - a + 1 ? r : b - a + 1;
        b -= d;
        r -= d;
        a -= (r + 1) / 2;  // b == a - 1 now: the question is longer
        b -= r / 2;
    }
    if (a < 0) a = 0;
    if (b < 0) b = 0;
}

// one 64-thread block per output row (question b, beam slot j)
__global__ void __launch_bounds__(64)
assemble_hop2_kernel(const long long* __restrict__ q_ids, const long long* __restrict__ q_mask, int Lq, const long long* __restrict__ doc_ids,
                     const int* __restrict__ arena_tokens, const long long* __restrict__ arena_offsets, const unsigned char* __restrict__ arena_empty,
                     long long n_docs, float* __restrict__ D1, int beam, int Lout, int bos, int eos, int pad, long long* __restrict__ out_ids,
                     long long* __restrict__ out_mask) {
    const int row = blockIdx.x, b = row / beam, lane = threadIdx.x;
    // question length incl. <s> and </s>: number of mask ones (right padded)
    int qn = 0;
    for (int p = lane; p < Lq; p += 64) qn += q_mask[(size_t)b * Lq + p] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
    int a = qn >= 2 ? qn - 2 : 0;  // raw question tokens
    long long doc = doc_ids[row];
    int dlen = 0;
    long long dbeg = 0;
    if (doc >= 0 && doc < n_docs) {
        dbeg = arena_offsets[doc];
        dlen = (int)(arena_offsets[doc + 1] - dbeg);
        if (arena_empty && arena_empty[doc] && D1 && lane == 0) D1[row] = -INFINITY;
    }
    int bk = dlen;
    longest_first(a, bk, a + bk + 4 - Lout);
    const int total = a + bk + 4;
    long long* oi = out_ids + (size_t)row * Lout;
    long long* om = out_mask + (size_t)row * Lout;
    for (int p = lane; p < Lout; p += 64) {
        long long t = pad;
        if (p == 0) t = bos;
        else if (p <= a) t = q_ids[(size_t)b * Lq + p];          // question tokens sit at 1..a in the hop-1 row
        else if (p == a + 1 || p == a + 2) t = eos;
        else if (p < a + 3 + bk) t = arena_tokens[dbeg + (p - a - 3)];
        else if (p == a + 3 + bk) t = eos;
        oi[p] = t;
        om[p] = p < total ? 1 : 0;
    }
}

}  // namespace
}  // namespace mdr

extern "C" int mdr_assemble_hop2(const int64_t* q_ids_dev, const int64_t* q_mask_dev, int batch, int q_len, const int64_t* doc_ids_dev, int beam,
                                 const int32_t* arena_tokens_dev, const int64_t* arena_offsets_dev, const uint8_t* arena_empty_dev, int64_t n_docs,
                                 float* hop1_scores_dev, int out_len, int bos_id, int eos_id, int pad_id, int64_t* out_ids_dev, int64_t* out_mask_dev,
                                 void* stream) {
    using namespace mdr;
    MDR_REQUIRE(batch >= 0 && beam >= 1 && q_len >= 2 && out_len >= 5, "bad shape batch=%d beam=%d q_len=%d out_len=%d", batch, beam, q_len, out_len);
    if (batch == 0) return MDR_OK;
    MDR_REQUIRE(q_ids_dev && q_mask_dev && doc_ids_dev && arena_tokens_dev && arena_offsets_dev && out_ids_dev && out_mask_dev, "NULL pointer");
    hipLaunchKernelGGL(assemble_hop2_kernel, dim3(batch * beam), dim3(64), 0, (hipStream_t)stream, (const long long*)q_ids_dev, (const long long*)q_mask_dev,
                       q_len, (const long long*)doc_ids_dev, (const int*)arena_tokens_dev, (const long long*)arena_offsets_dev,
                       (const unsigned char*)arena_empty_dev, (long long)n_docs, hop1_scores_dev, beam, out_len, bos_id, eos_id, pad_id,
                       (long long*)out_ids_dev, (long long*)out_mask_dev);
    MDR_HIP_TRY(hipGetLastError());
    return MDR_OK;
}

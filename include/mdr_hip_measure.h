/*
 * include/mdr_hip_measure.h -- hooks of MEASUREMENT builds of libmdrhip (variant libraries built next to the product one with
 *     python -m multihop_dense_retrieval_amd.build -DMDR_GEMM_ABL=5 --out=libmdrhip_gemm_timeline.so
 *     python -m multihop_dense_retrieval_amd.build -DMDR_I8_ABL=9  --out=libmdrhip_i8_timeline.so
 *     python -m multihop_dense_retrieval_amd.build -DMDR_ATTN_ABL=9 --out=libmdrhip_attn_timeline.so
 *     python -m multihop_dense_retrieval_amd.build -DMDR_CU_LANES=1 --out=libmdrhip_cu_lanes.so
 * and loaded by the measurement scripts through MDR_LIB_PATH). The product library (no -D) exports none of these, holds no
 * timeline globals and none of the ablation code paths: every MDR_*_ABL switch is a compile-time macro, because several of them
 * produce WRONG results by design (they remove one pipe's work to see what a kernel waits for).
 *
 * Replaces nothing in the reference (which has no native code); listed here so that the C ABI in mdr_hip.h stays the drop-in
 * surface only.
 */
#ifndef MDR_HIP_MEASURE_H
#define MDR_HIP_MEASURE_H
#ifdef __cplusplus
extern "C" {
#endif

/* -DMDR_GEMM_ABL=5 builds: the s_memtime timeline the persistent 256x256 GEMM accumulates (shader cycles of wave 0 summed over
 * workgroups: [0] wait + barrier A, [1..3] sub-phases 1-3, [4] wait + barrier B, [5] sub-phase 4, [6] epilogue, [7] K-tiles counted);
 * synchronises the device; reset != 0 clears it. Results of that build are correct. (scripts/measure/gpu_gemm_bench.py) */
int mdr_test_gemm_stamps(unsigned long long* out8_host, int reset);

/* -DMDR_I8_ABL=9 builds: the same kind of timeline for the 32-queries-per-wave int8 screen kernel ([0] wait + barrier,
 * [1] exchange + DMA issue, [2] MFMA chain, [3] epilogue, [4] bound sharing, [7] stages). (scripts/measure/gpu_i8_quick.py) */
int mdr_test_i8_stamps(unsigned long long* out8_host, int reset);

/* -DMDR_ATTN_ABL=9 builds: per-workgroup timeline of the LAST attention_stream_kernel launch, 8 words per workgroup
 * (linear id (z * B + b) * heads + h): wall_clock64 (100 MHz, chip-wide) at [0] entry, [1] first K/V chunk landed, [2] first query block
 * stored, [3] exit; [4] sequence length (0: the workgroup left at once), [5] HW_ID, [6] XCC_ID. Results of that build are correct.
 * (scripts/measure/gpu_attn_timeline.py) */
int mdr_test_attn_stamps(unsigned long long* out_host, int max_wgs);

/* -DMDR_CU_LANES=1 builds: CU-partitioned lanes (round 5; a measured negative for the headline loop: profiles/r05_cu_partitioned_lanes_negative.txt). The pipelined
 * loop runs two encoder forwards at once -- hop 2 of batch i (21 k tokens, persistent one-workgroup-per-CU GEMMs) beside hop 1 of batch i+1 (2.4 k tokens, ~140
 * short kernels) -- where the reference runs them one after the other (/root/reference/scripts/eval/eval_mhop_retrieval.py:148-150,168-171). A stream created here
 * may only use CUs [cu_lo, cu_hi) of the device (bit i of a CU mask = CU i / 8 of XCD i % 8 on MI355X: a range takes the same share of every XCD;
 * scripts/ubench/cu_mask_probe.hip); in that build mdr_encoder_forward sizes its persistent grids for the CUs of the stream it is given (the product build always
 * sizes them for the whole device). Results do not depend on the partition in exact arithmetic; the tile-shape choice moves with the CU count, so they are
 * not guaranteed bit-identical. (scripts/measure/cu_lanes.py, bench_loops.py --lane-cus) */
int mdr_stream_create_cu_range(int device, int cu_lo, int cu_hi, void** stream_out);
int mdr_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDR_HIP_MEASURE_H */

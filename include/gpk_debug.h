/* gpk_debug.h -- development aids of libgpk.so: NOT part of the drop-in boundary (include/gpk.h, include/gpk_mg.h).
 *
 * Kernel-variant switches for A/B measurements and for tests that force every instantiation, shader-clock stamps of the
 * diagonal-block kernels, probes that documented hardware behaviour, and the micro-benchmarks that fix
 * the roofline denominators bench.py quotes.  gpk_debug_set changes PROCESS-WIDE state (every handle of the process sees
 * it): tests toggle it and restore the default inside one function; the product path never calls it.
 */
#ifndef GPK_DEBUG_H
#define GPK_DEBUG_H

#include "gpk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* development aids (process-wide): key 0 = force the GEMM tile configuration (0 auto, 1 = 128x128, 2 = 64x64);
 * key 2 = run multi-RHS triangular solves as 4 column groups on concurrent streams (0 off, default); key 10 = 0: substitution
 * strips even when Dinv is given; key 12 = 0: SYRK then right-looking Cholesky on one stream instead of the two-partition
 * pipeline; key 13 = CUs of the chain partition (default 32); key 24 = workgroups per split-K product launch of the pipeline
 * (default 1000, 0 = no split); the full list is in tools/README.md */
int gpk_debug_set(int key, int value);
/* development aid: enable/disable and read the shader-clock phase stamps of the 64-wide diagonal-block kernels */
int gpk_debug_stamps(gpk_handle h, unsigned long long* host16, int enable);

/* ---- micro-benchmarks used to fix the roofline denominators ------------------------------------------------ */
int gpk_ubench_mfma_f64(gpk_handle h, int iters, double* host_tflops);      /* v_mfma_f64_16x16x4_f64 issue rate */
int gpk_ubench_hbm_write(gpk_handle h, size_t bytes, int iters, double* host_gbps);
int gpk_ubench_latency(gpk_handle h, int mode, double* host_cycles_per_op);
int gpk_ubench_xcc_map(gpk_handle h, int nblocks, int mode, int* host_out);   /* XCD id (HW_REG_XCC_ID) each workgroup ran on; mode 1: odd workgroups linger */   /* 0 dep. v_fma_f64, 1 indep. v_fma_f64, 2 dep. ds_read, 3 indep. ds_read, 4 dep. mfma_f64 (shader cycles per op, one wave) */
/* which CUs a stream created with hipExtStreamCreateWithCUMask(bits [first_bit, first_bit + nbits)) dispatches to: per
 * workgroup XCC_ID | HW_REG_HW_ID << 8 (tools/cu_mask_probe.py) */
int gpk_ubench_cu_census(gpk_handle h, int first_bit, int nbits, int nblocks, int* host_out);
/* development probe (tools/overlap_probe.py): C2 <- S^T S on a low-priority side stream while potrf(copy of H) runs on the
 * handle's stream; host_ms3 = {potrf alone, syrk alone, both concurrently}.  Round-1 finding: no overlap (5.5 vs 2.9 + 2.5 ms). */
int gpk_debug_overlap_probe(gpk_handle h, double* H, int n, int ldh, const double* S, int k, int lds, double* C2, int ldc,
                            double* host_ms3);

/* EXPERIMENT (round 3): plain NN product C = A B with the operand feed through LDS-DMA (global_load_lds) instead of global ->
 * VGPR -> LDS; same tile as the product kernel.  M, N multiples of 64, K of 16, even leading dimensions, 16-byte aligned A, B.
 * csrc/gpk_gemm_dma_probe.hip, tools/gemm_dma_probe.py. */
int gpk_debug_gemm_dma(gpk_handle h, int m, int n, int k, const double* A, int lda, const double* B, int ldb, double* C, int ldc);

#ifdef __cplusplus
}
#endif
#endif /* GPK_DEBUG_H */

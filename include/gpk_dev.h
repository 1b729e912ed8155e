/* gpk_dev.h -- entry points of libgpk_dev.so, the DEVELOPMENT build of the library (csrc/Makefile): the product sources compiled with
 * -DGPK_DEV plus csrc/dev/.  It exports everything libgpk.so exports (include/gpk.h, gpk_mg.h, gpk_debug.h) and, on top,
 *   - the look-ahead factorisation of round 5 (measured, not adopted: gpk_tune key 54) -- the superseded kernel designs of rounds 1-2
 *     (gpk_tune keys 5 = 0, 7 = 1, 21 = 0 / 2, 4 = 2) were removed in round 6,
 *   - the shader-clock stamps of the diagonal-block kernels,
 *   - probes that documented hardware behaviour and the micro-benchmarks that fix the roofline denominators.
 * tests/ and tools/ load it explicitly (gpk.Context(dev=True)); bench.py, the drivers and the host API never do.
 */
#ifndef GPK_DEV_H
#define GPK_DEV_H

#include "gpk.h"
#include "gpk_debug.h"

#ifdef __cplusplus
extern "C" {
#endif

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
#endif /* GPK_DEV_H */

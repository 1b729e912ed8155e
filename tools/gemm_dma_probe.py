#!/usr/bin/env python3
"""EXPERIMENT (verdict item 3 of round 2): operand feed of the fp64 GEMM through LDS-DMA (csrc/gpk_gemm_dma_probe.hip) against the product
kernel's global -> VGPR -> LDS staging, same 64 x 64 tile, plain NN products.  Checks the result, then times both."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0, dev=True)
lib = ctx.lib
def timed(fn, reps=5):
    fn(); fn(); ctx.synchronize(); best = 1e30
    for _ in range(reps):
        ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best
rng = np.random.RandomState(0)
# correctness on a small shape
m, n, k = 192, 256, 208
A = rng.normal(size=(m, k)); B = rng.normal(size=(k, n))
dA, dB, dC = ctx.array(A), ctx.array(B), ctx.empty(m, n)
assert lib.gpk_debug_gemm_dma(ctx.h, m, n, k, dA.ptr, dA.ld, dB.ptr, dB.ld, dC.ptr, dC.ld) == 0, lib.gpk_last_error(ctx.h)
err = np.max(np.abs(dC.download() - A @ B)) / np.max(np.abs(A @ B))
print('check %dx%dx%d: max rel err %.2e' % (m, n, k, err), flush=True)
assert err < 1e-13
for (m, n, k) in [(2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (2304, 4032, 6144), (4032, 512, 8400)]:
    A = ctx.empty(m, k); B = ctx.empty(k, n); C1 = ctx.empty(m, n); C2 = ctx.empty(m, n)
    A.upload(rng.normal(size=(m, k))); B.upload(rng.normal(size=(k, n)))
    lib.gpk_debug_set(0, 2); lib.gpk_debug_set(42, 0)                 # product kernel, 64 x 64 tile, one tile per workgroup
    t_ref = timed(lambda: ctx.gemm(0, 0, m, n, k, 1.0, A, B, 0.0, C1))
    lib.gpk_debug_set(0, 0); lib.gpk_debug_set(42, 1)
    t_best = timed(lambda: ctx.gemm(0, 0, m, n, k, 1.0, A, B, 0.0, C1))
    t_dma = timed(lambda: lib.gpk_debug_gemm_dma(ctx.h, m, n, k, A.ptr, A.ld, B.ptr, B.ld, C2.ptr, C2.ld))
    d = np.max(np.abs(C1.download() - C2.download()))
    fl = 2.0 * m * n * k / 1e9
    print('NN %dx%dx%d: product kernel 64x64 %.3f ms %.1f TF/s | product kernel auto config %.3f ms %.1f TF/s | LDS-DMA 64x64 %.3f ms %.1f TF/s   (max diff %.1e)'
          % (m, n, k, t_ref, fl / t_ref, t_best, fl / t_best, t_dma, fl / t_dma, d), flush=True)
    for a in (A, B, C1, C2): a.free()

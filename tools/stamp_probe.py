#!/usr/bin/env python3
"""Phase breakdown (shader-clock stamps) of the 64-wide diagonal-block kernels."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0, dev=True)
buf = (C.c_ulonglong * 16)()
ctx.lib.gpk_debug_stamps(ctx.h, None, 1)
rng = np.random.RandomState(0)
n = 64
M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n)
for rep in range(3):
    dA = ctx.array(A); info = ctx.potrf(dA); ctx.synchronize()
    L = ctx.array(np.linalg.cholesky(A)); B = ctx.array(rng.normal(size=(n, 4001)))
    ctx.trsm(L, B); ctx.synchronize()
    ctx.lib.gpk_debug_stamps(ctx.h, buf, 1)
    s = list(buf)
    print('potf2: load %d factor %d store %d | trsm<left>: zero %d stageL %d stageX %d solve %d store %d (cycles)' % (
        s[1] - s[0], s[2] - s[1], s[3] - s[2], s[5] - s[4], s[6] - s[5], s[7] - s[6], s[8] - s[7], s[9] - s[8]))
X = ctx.array(rng.normal(size=(4000, n)))
Ld = ctx.array(np.linalg.cholesky(A))
ctx.gemm(0, 1, 4000, n, n, 0.0, X, Ld, 1.0, X)   # no-op-ish warm
ctx.lib.gpk_debug_stamps(ctx.h, None, 0)
print('ubench mfma f64 (asm, 4 acc/wave, 4 waves/SIMD):', ctx.ubench_mfma_f64(20000), 'TF/s')

#!/usr/bin/env python3
"""Phase breakdown (shader-clock stamps of workgroup 0) of the fused 256-row strip TRSM kernel."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0, dev=True)
buf = (C.c_ulonglong * 16)()
ctx.lib.gpk_debug_stamps(ctx.h, None, 1)
rng = np.random.RandomState(0)
n = 256
M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n)
L = ctx.array(np.linalg.cholesky(A))
for rep in range(3):
    B = ctx.array(rng.normal(size=(n, 4001)))
    ctx.synchronize(); ctx.timer_start(); ctx.trsm(L, B); ms = ctx.timer_stop()
    ctx.lib.gpk_debug_stamps(ctx.h, buf, 1)
    s = list(buf)
    print('strip: stage %d solve0 %d step0 %d steps1-14 %d store %d total %d (clock64 ticks)  launch %.1f us' % (
        s[11] - s[10], s[12] - s[11], s[13] - s[12], s[14] - s[13], s[15] - s[14], s[15] - s[10], ms * 1e3))
ctx.lib.gpk_debug_stamps(ctx.h, None, 0)

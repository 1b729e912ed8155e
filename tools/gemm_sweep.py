#!/usr/bin/env python3
"""GEMM tile-configuration sweep for the shapes the TRSM/POTRF recursion produces."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
lib = ctx.lib
def timed(fn, reps=5):
    fn(); ctx.synchronize(); best = 1e30
    for _ in range(reps):
        ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best
N = 4001
print('shape (M x N x K)            ta tb   big ms   TF/s | small ms  TF/s')
for (ta, tb, shapes) in [(0, 0, [(64, N, 64), (128, N, 128), (256, N, 256), (512, N, 512), (1024, N, 1024), (2048, N, 2048), (4200, N, 4200)]),
                         (0, 1, [(8000, 448, 64), (4000, 256, 64), (2000, 64, 64), (7900, 7900, 512), (4000, 4000, 512), (2000, 2000, 512)]),
                         (1, 0, [(4001, 4001, 8400), (2000, 2000, 8400)])]:
    for (m, n, k) in shapes:
        A = ctx.empty(k, m) if ta else ctx.empty(m, k)
        B = ctx.empty(n, k) if tb else ctx.empty(k, n)
        Cm = ctx.empty(m, n)
        A.upload(np.random.normal(size=(A.rows, A.cols))); B.upload(np.random.normal(size=(B.rows, B.cols)))
        out = []
        for cfg in (1, 2):
            lib.gpk_debug_set(0, cfg)
            ms = timed(lambda: ctx.gemm(ta, tb, m, n, k, -1.0, A, B, 1.0, Cm))
            out.append((ms, 2.0 * m * n * k / ms / 1e9))
        lib.gpk_debug_set(0, 0)
        print(f'{m:5d} x {n:5d} x {k:5d}   {ta}  {tb}   {out[0][0]:7.3f} {out[0][1]:6.1f} | {out[1][0]:7.3f} {out[1][1]:6.1f}')
        A.free(); B.free(); Cm.free()

#!/usr/bin/env python3
"""128x128 vs 64x64 tile configuration on the GEMM shapes of the larger workloads (n10k, C5 shards)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
lib = ctx.lib
def timed(fn, reps=3):
    fn(); ctx.synchronize(); best = 1e30
    for _ in range(reps):
        ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best
for (ta, tb, m, n, k) in [(0, 0, 8192, 8192, 8192), (0, 0, 4200, 4001, 4200), (1, 0, 4001, 4001, 8400), (0, 0, 10500, 10001, 10500), (1, 0, 10001, 10001, 21000), (0, 1, 16000, 16000, 512), (0, 0, 5250, 10001, 5250), (0, 1, 8000, 8000, 512), (0, 0, 2048, 16001, 2048)]:
    A = ctx.empty(k, m) if ta else ctx.empty(m, k)
    B = ctx.empty(n, k) if tb else ctx.empty(k, n)
    Cm = ctx.empty(m, n)
    A.upload(np.random.normal(size=(A.rows, A.cols))); B.upload(np.random.normal(size=(B.rows, B.cols)))
    out = []
    for cfg in (2, 3, 4):
        lib.gpk_debug_set(0, cfg)
        ms = timed(lambda: ctx.gemm(ta, tb, m, n, k, -1.0, A, B, 1.0, Cm))
        out.append('%s %.3f ms %.1f TF/s' % ({1: '128^2', 2: '64^2', 3: '128x64/8w', 4: '128^2/16w'}[cfg], ms, 2.0 * m * n * k / ms / 1e9))
    lib.gpk_debug_set(0, 0)
    print('%s%s %dx%dx%d: %s' % ('T' if ta else 'N', 'T' if tb else 'N', m, n, k, ' | '.join(out)))
    A.free(); B.free(); Cm.free()

#!/usr/bin/env python3
"""Times the small-tile GEMM on three shapes; used with csrc/exp/libgpk_exp*.so (GPK_EXP ablations, wrong results by design)
copied over libgpk.so on the GPU box to attribute the gap to the MFMA issue ceiling (feed / LDS store / barrier)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
lib = ctx.lib
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def timed(fn, reps=5):
    fn(); ctx.synchronize(); best = 1e30
    for _ in range(reps):
        ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best
out = []
for (ta, tb, m, n, k) in [(0, 0, 4200, 4001, 4200), (1, 0, 4001, 4001, 8400), (0, 1, 4096, 4096, 512), (0, 0, 8192, 8192, 8192)]:
    A = ctx.empty(k, m) if ta else ctx.empty(m, k)
    B = ctx.empty(n, k) if tb else ctx.empty(k, n)
    Cm = ctx.empty(m, n)
    A.upload(np.random.normal(size=(A.rows, A.cols))); B.upload(np.random.normal(size=(B.rows, B.cols)))
    lib.gpk_debug_set(0, cfg)
    ms = timed(lambda: ctx.gemm(ta, tb, m, n, k, -1.0, A, B, 1.0, Cm))
    lib.gpk_debug_set(0, 0)
    out.append(f'{"TN"[ta]}{"TN"[tb] if False else ("T" if tb else "N")} {m}x{n}x{k}: {ms:.3f} ms {2.0*m*n*k/ms/1e9:.1f} TF/s')
    A.free(); B.free(); Cm.free()
print(' | '.join(out))

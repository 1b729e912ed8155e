#!/usr/bin/env python3
"""Tile-list (stream-K) launches vs one tile per workgroup on plain shapes: tile counts at and around whole rounds of resident
workgroups, forced tile configuration.  GPK env: SHAPES="ta,tb,m,n,k;..." overrides the list."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
lib = ctx.lib
def timed(fn, reps=5):
    fn(); fn(); ctx.synchronize(); best = 1e30
    for _ in range(reps):
        ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best
shapes = [(0, 0, 2048, 2048, 2048), (0, 0, 2048, 2112, 2048), (0, 0, 2048, 3072, 2048), (0, 0, 2048, 4096, 2048), (0, 0, 2048, 4160, 2048),
          (0, 0, 2304, 4032, 6144), (0, 0, 4096, 4096, 4096), (1, 0, 4032, 512, 8400), (0, 1, 4000, 512, 3500)]
if os.environ.get('SHAPES'):
    shapes = [tuple(int(v) for v in s.split(',')) for s in os.environ['SHAPES'].split(';')]
for (ta, tb, m, n, k) in shapes:
    A = ctx.empty(k, m) if ta else ctx.empty(m, k)
    B = ctx.empty(n, k) if tb else ctx.empty(k, n)
    Cm = ctx.empty(m, n)
    A.upload(np.random.normal(size=(A.rows, A.cols))); B.upload(np.random.normal(size=(B.rows, B.cols)))
    for cfg in (2, 3):
        out = []
        for sk in (0, 2):
            lib.gpk_debug_set(0, cfg); lib.gpk_debug_set(42, sk)
            ms = timed(lambda: ctx.gemm(ta, tb, m, n, k, -1.0, A, B, 1.0, Cm))
            out.append('sk=%d %.3f ms %.1f TF/s' % (sk, ms, 2.0 * m * n * k / ms / 1e9))
        t64 = ((m + 63) // 64) * ((n + 63) // 64)
        print('%s%s %dx%dx%d cfg %s (%d tiles of 64x64): %s' % ('T' if ta else 'N', 'T' if tb else 'N', m, n, k, {2: '64x64', 3: '128x64'}[cfg], t64, ' | '.join(out)), flush=True)
    lib.gpk_debug_set(0, 0); lib.gpk_debug_set(42, 1)
    A.free(); B.free(); Cm.free()

#!/usr/bin/env python3
"""Times the single-vector triangular solve (fused one-launch kernel vs. two launches per block) and checks it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(0)
for n in (4000, 4001, 8400, 100, 64, 1):
    M = rng.normal(size=(n, n)) / np.sqrt(n)
    Lh = np.tril(M) + 2.0 * np.eye(n)
    L = ctx.array(Lh)
    b = rng.normal(size=n)
    for trans in (0, 1):
        ref = np.linalg.solve(Lh.T if trans else Lh, b)
        out = []
        for fused in (1, 2, 0):
            ctx.lib.gpk_debug_set(4, fused)
            best = 1e9
            for rep in range(4):
                x = ctx.array(b)
                ctx.synchronize(); ctx.timer_start(); ctx.trsm(L, x, trans=bool(trans)); best = min(best, ctx.timer_stop())
            err = np.max(np.abs(x.download().ravel() - ref)) / np.max(np.abs(ref))
            out.append("%s %.1f us err %.1e" % ({1: "granules", 2: "flags", 0: "split"}[fused], best * 1e3, err))
        print('n=%5d trans=%d: %s' % (n, trans, ' | '.join(out)))
    L.free()
ctx.lib.gpk_debug_set(4, 1)

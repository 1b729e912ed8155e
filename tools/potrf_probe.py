#!/usr/bin/env python3
"""Cholesky timing: persistent outer-block kernel (gpk_debug_set(7,1)) vs one launch per 64-column panel (7,0)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0, dev=True)
rng = np.random.RandomState(0)
for n in (512, 1024, 4001):
    M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n)
    ref = np.linalg.cholesky(A)
    for mode in (1, 0):
        ctx.lib.gpk_debug_set(7, mode)
        best = 1e9
        for rep in range(4):
            dA = ctx.array(A)
            ctx.synchronize(); ctx.timer_start(); info = ctx.potrf(dA); best = min(best, ctx.timer_stop())
        err = np.max(np.abs(np.tril(dA.download()) - ref)) / np.max(np.abs(ref))
        print('n=%5d %s: %.1f us (%.1f us per 64-column panel) info %d err %.1e' % (n, 'persistent' if mode else 'per-panel ', best * 1e3, best * 1e3 / ((n + 63) // 64), info, err))
ctx.lib.gpk_debug_set(7, 1)
import ctypes as C
buf = (C.c_ulonglong * 16)()
ctx.lib.gpk_debug_stamps(ctx.h, None, 1)
n = 512
M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n)
for rep in range(3):
    dA = ctx.array(A); ctx.potrf(dA); ctx.synchronize()
    ctx.lib.gpk_debug_stamps(ctx.h, buf, 1)
    s = list(buf)
    print('WG1 (cycles): fetch+start %d | wait D0 %d | solve+store X %d | flag X + own update %d | potf2 %d | store+flag %d | total %d' % (
        s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[6] - s[5], s[7] - s[6], s[7] - s[0]))
ctx.lib.gpk_debug_stamps(ctx.h, None, 0)

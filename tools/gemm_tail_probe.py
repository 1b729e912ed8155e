#!/usr/bin/env python3
"""Lock-step vs staggered workgroups (gpk_debug_set key 15) on NN GEMM m x 4032 x 4352, dense and with leading zeros, measured
(a) one launch at a time from an idle chip and (b) sustained: 30 launches back to back (the regime inside a Gauss-Newton step).
Development probe."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(0)
n, k = 4032, 4352                                   # 63 column tiles
A = ctx.array(rng.normal(size=(5200, k))); B = ctx.array(rng.normal(size=(k, n))); Cm = ctx.empty(5200, n)
for stag in (0, 2):
    ctx.lib.gpk_debug_set(15, stag)
    print('stagger', stag)
    for m in (1024, 2048, 4160):
        tiles = ((m + 63) // 64) * 63
        best = 1e9
        for _ in range(4):
            ctx.timer_start(); ctx.gemm(0, 0, m, n, k, -1.0, A, B, 1.0, Cm); best = min(best, ctx.timer_stop())
        ctx.timer_start()
        for _ in range(30):
            ctx.gemm(0, 0, m, n, k, -1.0, A, B, 1.0, Cm)
        sus = ctx.timer_stop() / 30
        print(f'  m={m:5d} tiles={tiles:5d} waves={tiles / 1024:5.2f}  single {best * 1e3:8.1f} us {2.0 * m * n * k / best / 1e9:6.1f} TF/s |'
              f' sustained {sus * 1e3:8.1f} us {2.0 * m * n * k / sus / 1e9:6.1f} TF/s')
ctx.lib.gpk_debug_set(15, 0)

#!/usr/bin/env python3
"""gpk_potrf at a few orders: default (right-looking, one stream) vs the two-partition pipeline (gpk_debug_set(20, max_n))."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
from src.sample_points import sampled_pts_rdm
ctx = gpk.Context(0)
for Nd, Nb in ((2000, 200), (4000, 400), (10000, 1000)):
    np.random.seed(0)
    Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
    N = 2 * Nd + Nb
    T = ctx.empty(N, N)
    for mode in (0, 1):
        ctx.lib.gpk_debug_set(20, 1000000 if mode else 0)
        best = 1e9
        for rep in range(3):
            ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive', out=T)
            ctx.timer_start(); info = ctx.potrf(T); ms = ctx.timer_stop(); best = min(best, ms)
        print(f'N={N} pipelined={mode}: {best:.2f} ms, info {info}, {N**3/3/best/1e9:.1f} TF/s', flush=True)
    ctx.lib.gpk_debug_set(20, 0)
    T.free()

#!/usr/bin/env python3
"""Plain Cholesky at the orders of the north-star size (H: 10001, Theta: 21000): one stream (default) vs the two-partition pipeline with
chain partitions of 32 / 64 / 96 CUs (gpk_debug_set 13, 20)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
from src.sample_points import sampled_pts_rdm
ctx = gpk.Context(0)
for Nd, Nb in ((5000, 1), (10000, 1000)):
    np.random.seed(0); Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]])); N = 2 * Nd + Nb
    T = ctx.empty(N, N)
    for cus in (0, 32, 64, 96, 128):
        ctx.lib.gpk_debug_set(20, 1000000 if cus else 0)
        if cus:
            ctx.lib.gpk_debug_set(13, cus)
        best = 1e9
        for r in range(3):
            ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive', out=T)
            ctx.timer_start(); info = ctx.potrf(T); best = min(best, ctx.timer_stop())
        print(f'N={N} chain partition {cus or "none (one stream)"}: {best:.2f} ms', flush=True)
    ctx.lib.gpk_debug_set(20, 0); ctx.lib.gpk_debug_set(13, 32)
    T.free()

#!/usr/bin/env python3
"""Outer block width of the right-looking Cholesky factorisation (gpk_debug_set key 51) at the orders of configs 2 / north-star / 5."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
from src.sample_points import sampled_pts_rdm
ctx = gpk.Context(0)
for Nd, Nb in ((2000, 1), (4000, 400), (5000, 1), (10000, 1000), (16000, 2000)):
    np.random.seed(0); Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]])); N = 2 * Nd + Nb
    T = ctx.empty(N, N)
    out = []
    for ob in (256, 384, 512, 768, 1024):
        ctx.lib.gpk_debug_set(51, ob)
        best = 1e9
        for r in range(3):
            ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive', out=T)
            ctx.timer_start(); info = ctx.potrf(T); best = min(best, ctx.timer_stop())
        out.append(f'{ob}: {best:.2f}')
    print(f'N={N}  ' + '  '.join(out) + ' ms', flush=True)
    ctx.lib.gpk_debug_set(51, 512)
    T.free()

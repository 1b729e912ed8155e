#!/usr/bin/env python3
"""Cholesky of Theta at the n10k workload (N = 21000, nugget 1e-13) under different kernel configurations: is a reported
non-positive pivot a rounding-order effect of the marginally positive-definite matrix or a kernel bug?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
from src.sample_points import sampled_pts_rdm
ctx = gpk.Context(0, dev=True)
np.random.seed(0)
Xd, Xb = sampled_pts_rdm(10000, 1000, [[0, 1], [0, 1]], time_dependent=False)
for nug in (1e-13, 1e-6):
    for cfg, name in [(1, 'fused panel'), (0, 'potf2 + trsm launches')]:
        ctx.lib.gpk_debug_set(5, cfg)
        T, ratios = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, nug, 'adaptive')
        ctx.synchronize(); ctx.timer_start(); info = ctx.potrf(T); ms = ctx.timer_stop()
        print('nugget %.0e %-22s info %d  %.1f ms' % (nug, name, info, ms))
        T.free()
ctx.lib.gpk_debug_set(5, 1)

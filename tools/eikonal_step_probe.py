#!/usr/bin/env python3
"""Time of one Gauss-Newton step of the Eikonal system (N_domain = 2000, N_boundary = 400: Theta of order 8400, 6000 unknowns):
leading-zero layout (regrouped unknowns, default) vs dense schedule (gpk_debug_set(23, 0)).  Development probe."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(1)
Nd, Nb = 2000, 400
Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
T, _ = ctx.assemble('Eikonal', 'Gaussian', 0.2, Xd, Xb, 1e-5, 'adaptive')
assert ctx.potrf(T) == 0
for mode in (1, 2, 0, 1, 2, 0):                                 # exact profile, closed-form staircase, dense
    ctx.lib.gpk_debug_set(23, mode)
    prob = gpk.GNProblem(ctx, 'Eikonal', Nd, Nb, np.ones(Nd), np.zeros(Nb), T, p0=0.1)
    z = ctx.array(np.zeros(3 * Nd))
    ctx.gn_step(prob, z); ctx.synchronize()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(8):
        ctx.gn_step(prob, z)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 8
    pr = ctx.prof_read(); ctx.prof_enable(False)
    print(f"{ {1: 'leading-zero, exact profile ', 2: 'leading-zero, closed form   ', 0: 'dense schedule              '}[mode] }: {dt * 1e3:7.2f} ms per step  (solve {pr['trsm_ms'] / 8:.2f}, product + factorisation {pr['syrk_ms'] / 8:.2f}, trsv {pr['trsv_update_ms'] / 8:.2f})")
ctx.lib.gpk_debug_set(23, 1)

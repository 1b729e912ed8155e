#!/usr/bin/env python3
"""A/B of the GEMM tile configuration on the three largest launches of a C2 Gauss-Newton step (per-phase event timers)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')); sys.path.insert(0, ROOT)
import gpk, bench
ctx = gpk.Context(0)
Nd, Nb = 4000, 400
Xd, Xb, f, g, z0 = bench.synthetic_problem(Nd, Nb)
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-13, 'adaptive')
assert ctx.potrf(T) == 0
prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
for cfg in (0, 1, 2, 0, 1):
    ctx.lib.gpk_debug_set(0, cfg)
    z = ctx.array(z0)
    ctx.gn_step(prob, z)
    ctx.prof_enable(True)
    for _ in range(4): ctx.gn_step(prob, z)
    p = ctx.prof_read(); ctx.prof_enable(False)
    print('cfg', cfg, {k: round(v / p['steps'], 3) for k, v in p.items() if k != 'steps'})

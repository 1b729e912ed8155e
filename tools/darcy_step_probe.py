"""Darcy Gauss-Newton step time, leading-zero layout (default) against the dense schedule (gpk_tune key 23 = 0), over problem sizes:
the leading-zero path issues its products and the factorisation of H on one stream (no two-partition pipeline), the dense path
pipelines them for orders <= 7000 -- this probe shows where each wins."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(1)
for Nd, Nb, Ndata in ((200, 60, 20), (400, 100, 60), (700, 120, 60), (1000, 160, 60), (1166, 180, 60), (1600, 200, 60)):
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    Tu, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-8, 'adaptive')
    Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-8, 'adaptive')
    assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
    data = 0.05 * rng.normal(size=Ndata)
    z0 = 0.3 * rng.normal(size=6 * Nd)
    out = []
    for mode in (1, 0):
        ctx.lib.gpk_debug_set(23, mode)
        prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, np.ones(Nd), np.zeros(Nb), Tu, p0=1e-3, data_u=data, L2=Ta)
        z = ctx.array(z0)
        for _ in range(2):
            ctx.gn_step(prob, z)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(6):
            ctx.gn_step(prob, z)
        ctx.synchronize(); out.append((time.perf_counter() - t0) / 6 * 1e3)
        prob.release_workspace()
    ctx.lib.gpk_debug_set(23, 1)
    print(f'N_d {Nd:5d}  n_z+1 {6 * Nd + 1:5d}: leading-zero {out[0]:7.3f} ms   dense {out[1]:7.3f} ms', flush=True)

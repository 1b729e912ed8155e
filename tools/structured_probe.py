#!/usr/bin/env python3
"""Optional structured solve (GNProblem(structured=True)) at BASELINE config 2 (nugget 1e-13): step time, phases, and the iterates
against the default path (triangular solve every step) over 8 steps from the same start."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')); sys.path.insert(0, ROOT)
import gpk
from oracle import gp_oracle as O
from src.sample_points import sampled_pts_rdm
Nd, Nb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 400)
np.random.seed(1)
Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
truth = O.elliptic_truth(Xd[:, 0], Xd[:, 1])
init = np.random.normal(0.0, 1.0, Nd)
ctx = gpk.Context(0)
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-13, 'adaptive')
assert ctx.potrf(T) == 0
res = {}
for structured in (False, True, 2):
    t0 = time.perf_counter()
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, structured=structured)
    ctx.synchronize(); tprep = time.perf_counter() - t0
    z = ctx.array(init); its = []
    for k in range(10):
        ctx.synchronize(); t0 = time.perf_counter()
        loss, info = ctx.gn_step(prob, z, 1.0); assert info == 0
        dt = time.perf_counter() - t0
        its.append((z.download().ravel().copy(), loss, dt))
    res[structured] = its
    print({False: 'default   ', True: 'structured', 2: 'gram level'}[structured], 'setup %.1f ms;' % (tprep * 1e3), 'ms per step (last 6): %.3f;' % (1e3 * np.mean([i[2] for i in its[4:]])),
          'rms error of the last iterate %.3e' % np.sqrt(np.mean((its[-1][0] - truth) ** 2)))
    prob.release_workspace()
for mode in (True, 2):
    print(mode, 'relative difference of the iterates, step by step:', ' '.join('%.1e' % (np.linalg.norm(a[0] - b[0]) / np.linalg.norm(b[0])) for a, b in zip(res[mode], res[False])))
    print(mode, 'relative difference of the in-step losses      :', ' '.join('%.1e' % (abs(a[1] - b[1]) / abs(b[1])) for a, b in zip(res[mode], res[False])))

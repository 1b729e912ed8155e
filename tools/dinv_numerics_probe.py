#!/usr/bin/env python3
"""Gauss-Newton iterates at BASELINE config 2 (nugget 1e-13) with the solve phase through inverted diagonal blocks of 256 .. 2048 rows
against the substitution path (gpk_debug_set(10, 0)): relative difference of the iterates after each of 6 steps."""
import os, sys
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
init = np.random.normal(0.0, 1.0, Nd)
ctx = gpk.Context(0)
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-13, 'adaptive')
assert ctx.potrf(T) == 0
def run(block, use):
    ctx.lib.gpk_debug_set(10, use)
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, dinv=block)
    z = ctx.array(init); out = []
    for _ in range(6):
        loss, info = ctx.gn_step(prob, z, 1.0); assert info == 0
        out.append(z.download().ravel().copy())
    ctx.lib.gpk_debug_set(10, 1)
    return out
ref = run(256, 0)
truth = O.elliptic_truth(Xd[:, 0], Xd[:, 1])
print('substitution: rms error of the last iterate', np.sqrt(np.mean((ref[-1] - truth) ** 2)))
for b in (256, 512, 1024, 2048):
    got = run(b, 1)
    print(b, ' '.join('%.1e' % (np.linalg.norm(a - r) / np.linalg.norm(r)) for a, r in zip(got, ref)), ' rms error', np.sqrt(np.mean((got[-1] - truth) ** 2)))

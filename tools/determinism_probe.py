#!/usr/bin/env python3
"""Repeats the same 6 Gauss-Newton steps at BASELINE config 2 from the same start 30 times and compares the iterates BITWISE: the
split-K reductions (last-arriver sums in chunk order), the two-partition pipeline and the fused triangular solves must not make the
result depend on the order in which workgroups happen to arrive."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')); sys.path.insert(0, ROOT)
import gpk
from oracle import gp_oracle as O
from src.sample_points import sampled_pts_rdm
Nd, Nb = 4000, 400
np.random.seed(2)
Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
init = np.random.normal(0.0, 1.0, Nd)
ctx = gpk.Context(0)
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-13, 'adaptive')
assert ctx.potrf(T) == 0
prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
ref = None; bad = 0
for rep in range(30):
    z = ctx.array(init)
    losses = []
    for _ in range(6):
        loss, info = ctx.gn_step(prob, z, 1.0); assert info == 0
        losses.append(loss)
    out = (z.download().ravel().copy(), np.array(losses))
    if ref is None: ref = out
    elif not (np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])):
        bad += 1; print('repeat', rep, 'differs: max |dz| =', np.max(np.abs(out[0] - ref[0])))
    z.free()
print('30 repeats of 6 steps:', 'bit-identical' if bad == 0 else f'{bad} repeats differ')

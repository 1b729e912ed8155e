import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/nonlinpdes-gpsolver_amd')
import numpy as np, gpk
from oracle import gp_oracle as O
ctx = gpk.Context(0)
rng = np.random.RandomState(12)
Nd, Nb = 650, 99
Xd = np.stack([rng.uniform(0, 1, Nd), rng.uniform(-1, 1, Nd)], axis=1)
Xb = np.stack([rng.uniform(0, 1, Nb), rng.uniform(-1, 1, Nb)], axis=1)
f = np.zeros(Nd); g = -np.sin(np.pi * Xb[:, 1]) * (rng.uniform(size=Nb) < 0.4)
T, _ = ctx.assemble('Burgers', 'anisotropic_Gaussian', [0.3, 0.05], Xd, Xb, 1e-5, 'adaptive')
assert ctx.potrf(T) == 0
L = np.tril(T.download())
z0 = rng.normal(size=3 * Nd)
sysm = O.BurgersSystem(1.0, 0.02, f, g)
sol_ref, hist_ref = O.gn_method(sysm, [L], z0, 1, 1)
for mode in (1, 0):
    ctx.lib.gpk_debug_set(23, mode)
    prob = gpk.GNProblem(ctx, 'Burgers', Nd, Nb, f, g, T, p0=1.0, p1=0.02)
    z = ctx.array(z0)
    loss, info = ctx.gn_step(prob, z)
    zz = z.download()
    print('mode', mode, 'loss', loss, hist_ref[0], 'rel diff vs oracle after 1 step', np.linalg.norm(zz - sol_ref) / np.linalg.norm(sol_ref), 'info', info)
    _, _, delta, _ = prob.workspace()
    d = delta.download()
    dref = z0 - sol_ref
    bad = np.argsort(-np.abs(d - dref))[:10]
    print('  worst unknown indices', bad, (d - dref)[bad])

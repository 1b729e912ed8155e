"""Conditioning of the Gauss-Newton matrix H of every system at mid size (GPU box; numpy eigvalsh on the host): decides whether the
look-ahead factorisation of H may multiply by explicit inverses of its 512-column diagonal blocks (round 4)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
import bench

ctx = gpk.Context(0)


def report(name, H):
    w = np.linalg.eigvalsh(H)
    worst = 0.0
    for k0 in range(0, H.shape[0], 512):
        wk = np.linalg.eigvalsh(H[k0:k0 + 512, k0:k0 + 512])
        worst = max(worst, wk[-1] / wk[0])
    print(f'{name}: order {H.shape[0]}, eig(H) in [{w[0]:.3e}, {w[-1]:.3e}], cond(H) = {w[-1] / w[0]:.3e}; worst 512-block cond = {worst:.3e}', flush=True)


for wl in ('c3', 'c4'):
    P = bench.system_problem(wl)
    Xd, Xb = P['Xd'], P['Xb']
    fac = []
    for lay in P['layouts']:
        T, _ = ctx.assemble(lay, P['kernel'], P['kp'], Xd, Xb, P['nugget'], 'adaptive')
        assert ctx.potrf(T) == 0
        fac.append(T)
    prob = gpk.GNProblem(ctx, P['system'], Xd.shape[0], Xb.shape[0], P['f'], P['g'], fac[0], p0=P['p0'], p1=P['p1'], data_u=P['data'],
                         L2=fac[1] if len(fac) > 1 else None)
    z = ctx.array(P['z0'])
    for it in range(9):
        if it in (0, 3, 8):
            H, _ = ctx.gn_hessian_grad(prob, z)
            report(f'{wl} step {it}', H)
        ctx.gn_step(prob, z)
Xd, Xb, f, g, z0 = bench.synthetic_problem(4000, 400)
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-13, 'adaptive')
assert ctx.potrf(T) == 0
prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', 4000, 400, f, g, T, p0=1.0, p1=3.0)
z = ctx.array(z0)
for it in range(5):
    if it in (0, 2, 4):
        H, _ = ctx.gn_hessian_grad(prob, z)
        report(f'c2 step {it}', H)
    ctx.gn_step(prob, z)

#%%
"""Burgers equation u_t + alpha u u_x - nu u_xx = 0 on (t,x) in [0,1]x[-1,1] with the GP solver on an MI355X.
Same command line as the reference's main_Burgers1d.py (flags, defaults, --randomseed)."""
import argparse

import numpy as onp
from numpy import random

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, figures_enabled
from src.solver import solver_GP


def get_parser():
    parser = argparse.ArgumentParser(description='Burgers equation GP solver')
    parser.add_argument("--alpha", type=float, default=1.0)
    parser.add_argument("--nu", type=float, default=0.02)
    add_kernel_and_sampling(parser, 'anisotropic_Gaussian', [0.3, 0.05], 1e-5, 1000, 200, kp_nargs='+')
    add_gn_and_logs(parser, 'rdm', 8)
    parser.add_argument("--randomseed", type=int, default=0)
    return parser.parse_args()


cfg = get_parser()
random.seed(cfg.randomseed)
print(f"[Seeds] random seeds: {cfg.randomseed}")
show = figures_enabled(cfg)

##### step 0: initialize the solver
alpha, nu = cfg.alpha, cfg.nu
solver = solver_GP(cfg, PDE_type="Burgers")


##### step 1: set the equation, rhs, bdy
def u(x1, x2):          # initial condition at t = 0, zero on the lateral boundaries
    return -onp.sin(onp.pi * x2) * (x1 == 0) + 0 * (x2 == 0)


def f(x1, x2):
    return 0


solver.set_equation(bdy=u, rhs=f, domain=onp.array([[0, 1], [-1, 1]]))

##### step 2: sample points
solver.auto_sample(cfg.N_domain, cfg.N_boundary, sampled_type=cfg.sampled_type)
if show:
    solver.show_sample()

###### step 3: solve the equation using GP + GN iterations
solver.solve()
if show:
    solver.show_loss_hist()

##### step 4: error calculation on test points: Cole-Hopf solution by 80-point Gauss-Hermite quadrature
Gauss_pts, weights = onp.polynomial.hermite.hermgauss(80)


def u_truth(x1, x2):
    y = x2[..., None] - onp.sqrt(4 * nu * x1[..., None]) * Gauss_pts
    e = weights * onp.exp(-onp.cos(onp.pi * y) / (2 * onp.pi * nu))
    return -onp.sum(onp.sin(onp.pi * y) * e, axis=-1) / onp.sum(e, axis=-1)


N_pts = 60
xx = onp.linspace(0, 1, N_pts)
yy = onp.linspace(-1, 1, N_pts)
XX, YY = onp.meshgrid(xx, yy)
X_test = onp.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
test_truth = u_truth(X_test[:, 0], X_test[:, 1])
solver.test(X_test)
solver.get_test_error(test_truth)
if show:
    solver.contour_of_test_err(XX, YY)

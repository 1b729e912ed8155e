#%%
"""Nonlinear elliptic equation -Delta u + alpha*u^m = f on [0,1]^2 with the GP solver on an MI355X.
Same command line as the reference's main_NonLinElliptic2d.py (README.md:15 of the reference):
    python main_NonLinElliptic2d.py --kernel Gaussian --kernel_parameter 0.2 --nugget 1e-13 --N_domain 900 --N_boundary 124 --GNsteps 4
The manufactured solution's right-hand side is written out analytically (the reference differentiates u with jax.grad)."""
import argparse

import numpy as onp

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, figures_enabled
from src.solver import solver_GP


def get_parser():
    parser = argparse.ArgumentParser(description='NonLinElliptic equation GP solver')
    parser.add_argument("--alpha", type=float, default=1.0)
    parser.add_argument("--m", type=float, default=3.0)
    add_kernel_and_sampling(parser, 'Gaussian', 0.2, 1e-13, 900, 124)
    parser.add_argument("--pen_lambda", type=float, default=1e-10)      # for the relaxation approach
    add_gn_and_logs(parser, 'rdm', 4, method_choices=['elimination', 'relaxation'])
    return parser.parse_args()


cfg = get_parser()
show = figures_enabled(cfg)

##### step 0: initialize the solver
solver = solver_GP(cfg, PDE_type="Nonlinear_elliptic")

##### step 1: set the equation, rhs, bdy
alpha, m = cfg.alpha, cfg.m
pi = onp.pi


def u(x1, x2):
    return onp.sin(pi * x1) * onp.sin(pi * x2) + 2 * onp.sin(4 * pi * x1) * onp.sin(4 * pi * x2)


def f(x1, x2):          # -Laplace(u) + alpha*u^m
    return 2 * pi ** 2 * onp.sin(pi * x1) * onp.sin(pi * x2) + 64 * pi ** 2 * onp.sin(4 * pi * x1) * onp.sin(4 * pi * x2) \
        + alpha * (u(x1, x2) ** m)


solver.set_equation(bdy=u, rhs=f, domain=onp.array([[0, 1], [0, 1]]), print_option=cfg.print_hist)

##### step 2: sample points
solver.auto_sample(cfg.N_domain, cfg.N_boundary, sampled_type=cfg.sampled_type, print_option=cfg.print_hist)
if show:
    solver.show_sample()

##### step 3: solve the equation using GP + GN iterations
solver.solve(method=cfg.method, pen_lambda=cfg.pen_lambda, print_option=cfg.print_hist)
if show:
    solver.show_loss_hist()

##### step 4: error calculation on training points
pts_truth = u(solver.eqn.X_domain[:, 0], solver.eqn.X_domain[:, 1])
solver.collocation_pts_err(pts_truth)

##### step 5: error calculation on test points
N_pts = 60
xx = onp.linspace(0, 1, N_pts)
yy = onp.linspace(0, 1, N_pts)
XX, YY = onp.meshgrid(xx, yy)
X_test = onp.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
test_truth = u(X_test[:, 0], X_test[:, 1])
solver.test(X_test)
solver.get_test_error(test_truth)
if show:
    solver.contour_of_test_err(XX, YY)

#%%
"""Regularised Eikonal equation |grad u|^2 = f + eps*Delta u on [0,1]^2 with the GP solver on an MI355X.
Same command line as the reference's main_Eikonal2d.py."""
import argparse

import numpy as onp

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, figures_enabled
from reference_solver.Cole_Hopf_for_Eikonal import solve_Eikonal
from src.solver import solver_GP


def get_parser():
    parser = argparse.ArgumentParser(description='Eikonal equation GP solver')
    parser.add_argument("--eps", type=float, default=1e-1)
    add_kernel_and_sampling(parser, 'Gaussian', 0.2, 1e-5, 1000, 200)
    add_gn_and_logs(parser, 'zero', 8)
    return parser.parse_args()


cfg = get_parser()
show = figures_enabled(cfg)

##### step 0: initialize the solver
solver = solver_GP(cfg, PDE_type="Eikonal")


###### step 1: set the equation, rhs, bdy
def u(x1, x2):
    return 0


def f(x1, x2):
    return 1


solver.set_equation(bdy=u, rhs=f, domain=onp.array([[0, 1], [0, 1]]))

##### step 2: sample points
solver.auto_sample(cfg.N_domain, cfg.N_boundary, sampled_type=cfg.sampled_type)
if show:
    solver.show_sample()

##### step 3: solve the equation using GP + GN iterations
solver.solve()
if show:
    solver.show_loss_hist()

##### step 4: error calculation on the interior of a 60x60 grid, truth by Cole-Hopf + finite differences
N_pts = 60
xx = onp.linspace(0, 1, N_pts)[1:-1]
yy = onp.linspace(0, 1, N_pts)[1:-1]
XX, YY = onp.meshgrid(xx, yy)
X_test = onp.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
solver.test(X_test)
XX, YY, test_truth = solve_Eikonal(N_pts - 2, cfg.eps)
solver.get_test_error(test_truth.flatten())
if show:
    solver.contour_of_test_err(XX, YY)

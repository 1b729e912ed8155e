#%%
"""Darcy-flow inverse problem -div(a grad u) = f: recover a from noisy observations of u, GP solver on an MI355X.
Same command line as the reference's main_DarcyFlow2d.py (flags, defaults, --randomseed)."""
import argparse

import numpy as onp
from numpy import random
from scipy.interpolate import griddata

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, figures_enabled
from reference_solver.FD_for_Darcy_flow import FD_Darcy_flow_2d
from src.solver import solver_GP


def get_parser():
    parser = argparse.ArgumentParser(description='Darcy Flow GP solver')
    add_kernel_and_sampling(parser, 'Gaussian', 0.2, 1e-8, 400, 100)
    parser.add_argument("--N_data", type=int, default=60)
    parser.add_argument("--noise_level", type=float, default=1e-3)
    add_gn_and_logs(parser, 'rdm', 8)
    parser.add_argument("--randomseed", type=int, default=9999)
    return parser.parse_args()


cfg = get_parser()
random.seed(cfg.randomseed)
print(f"[Seeds] random seeds: {cfg.randomseed}")
show = figures_enabled(cfg)

###### step 0: initialize the solver
solver = solver_GP(cfg, PDE_type="Darcy_flow2d")


###### step 1: set the equation, rhs, bdy
def u(x1, x2):
    return 0


def f(x1, x2):
    return 1


solver.set_equation(bdy=u, rhs=f, domain=onp.array([[0, 1], [0, 1]]))

# step 2: sample points
solver.auto_sample_IP(cfg.N_domain, cfg.N_boundary, cfg.N_data, sampled_type=cfg.sampled_type)
if show:
    solver.show_sample_IP()

###### step 3: observations = finite-difference solution on an 80x80 grid, linearly interpolated to the data points
N_pts_per_dim = 80
xx = onp.linspace(0, 1, N_pts_per_dim)
yy = onp.linspace(0, 1, N_pts_per_dim)
XX, YY = onp.meshgrid(xx, yy)
XXv, YYv = XX.flatten(), YY.flatten()


def a(x1, x2):          # true permeability
    s = onp.sin(2 * onp.pi * x1) + onp.sin(2 * onp.pi * x2)
    return onp.exp(s) + onp.exp(-s)


u_truth_grid = FD_Darcy_flow_2d(N_pts_per_dim - 2, a, f)
data_u = griddata((XXv, YYv), u_truth_grid.reshape(-1, 1), (solver.eqn.X_data[:, 0], solver.eqn.X_data[:, 1]), method='linear')[:, 0]
solver.get_observed_data(data_u, cfg.noise_level)

##### step 4: solve the equation using GP + GN iterations
solver.solve()
if show:
    solver.show_loss_hist()

##### step 5: GP interpolation and test accuracy on the FD grid
X_test = onp.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
solver.test(X_test)
test_u = onp.reshape(solver.eqn.extended_sol_u, (N_pts_per_dim, N_pts_per_dim))
test_a = onp.reshape(solver.eqn.extended_sol_a, (N_pts_per_dim, N_pts_per_dim))
test_truth_u = u_truth_grid
test_truth_a = a(XX, YY)
err_u = onp.sqrt(onp.mean((test_u - test_truth_u) ** 2))
err_a = onp.sqrt(onp.mean((onp.exp(test_a) - test_truth_a) ** 2))
print(f'[Test error] u: L2 error {err_u}, max error {onp.max(abs(test_u - test_truth_u))}')
print(f'[Test error] a: L2 error {err_a}, max error {onp.max(abs(onp.exp(test_a) - test_truth_a))}')

if show:
    import matplotlib.pyplot as plt
    fig = plt.figure()
    for k, (Z, title) in enumerate([(test_truth_a, 'Truth a(x)'), (onp.exp(test_a), 'Recovered a(x)'),
                                    (test_truth_u, 'Truth u(x)'), (test_u, 'Recovered u(x)')]):
        ax = fig.add_subplot(2, 2, k + 1)
        cs = ax.contourf(XX, YY, Z, 50, cmap=plt.cm.coolwarm)
        ax.set_xlabel('x_1'); ax.set_ylabel('x_2'); ax.set_title(title)
        fig.colorbar(cs)
    fig.tight_layout()
    plt.show()

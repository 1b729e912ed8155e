#%%
"""Darcy-flow inverse problem -div(a grad u) = f: recover a from noisy observations of u, GP solver on an MI355X.
Same command line as the reference's main_DarcyFlow2d.py (flags, defaults, --randomseed)."""
import argparse

import numpy as onp
from scipy.interpolate import griddata

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, figures_enabled, seed_from, tensor_grid
from reference_solver.FD_for_Darcy_flow import FD_Darcy_flow_2d
from src.solver import solver_GP

UNIT_SQUARE = [[0, 1], [0, 1]]
GRID = 80                                    # finite-difference / test grid per dimension


def parse(argv=None):
    parser = argparse.ArgumentParser(description='Darcy Flow GP solver')
    add_kernel_and_sampling(parser, 'Gaussian', 0.2, 1e-8, 400, 100)
    parser.add_argument("--N_data", type=int, default=60)
    parser.add_argument("--noise_level", type=float, default=1e-3)
    add_gn_and_logs(parser, 'rdm', 8)
    parser.add_argument("--randomseed", type=int, default=9999)
    return parser.parse_args(argv)


def permeability(x1, x2):
    s = onp.sin(2 * onp.pi * x1) + onp.sin(2 * onp.pi * x2)
    return onp.exp(s) + onp.exp(-s)


def source(x1, x2):
    return 1


def rms(a):
    return onp.sqrt(onp.mean(a ** 2))


def main(argv=None):
    cfg = parse(argv)
    seed_from(cfg)
    show = figures_enabled(cfg)
    solver = solver_GP(cfg, PDE_type="Darcy_flow2d")
    solver.set_equation(bdy=lambda x1, x2: 0, rhs=source, domain=onp.array(UNIT_SQUARE))
    solver.auto_sample_IP(cfg.N_domain, cfg.N_boundary, cfg.N_data, sampled_type=cfg.sampled_type)
    if show:
        solver.show_sample_IP()

    # observations: finite-difference solution on the grid, linearly interpolated to the data points, plus noise
    XX, YY, X_grid = tensor_grid(GRID, *UNIT_SQUARE)
    u_grid = FD_Darcy_flow_2d(GRID - 2, permeability, source)
    Xo = solver.eqn.X_data
    observed = griddata((XX.flatten(), YY.flatten()), u_grid.reshape(-1, 1), (Xo[:, 0], Xo[:, 1]), method='linear')[:, 0]
    solver.get_observed_data(observed, cfg.noise_level)

    solver.solve()
    if show:
        solver.show_loss_hist()

    # GP interpolation of both fields on the grid (the unknown is log a)
    solver.test(X_grid)
    u_gp = onp.reshape(solver.eqn.extended_sol_u, (GRID, GRID))
    a_gp = onp.exp(onp.reshape(solver.eqn.extended_sol_a, (GRID, GRID)))
    a_true = permeability(XX, YY)
    print(f'[Test error] u: L2 error {rms(u_gp - u_grid)}, max error {onp.max(abs(u_gp - u_grid))}')
    print(f'[Test error] a: L2 error {rms(a_gp - a_true)}, max error {onp.max(abs(a_gp - a_true))}')

    if show:
        import matplotlib.pyplot as plt
        fig = plt.figure()
        for k, (Z, title) in enumerate([(a_true, 'Truth a(x)'), (a_gp, 'Recovered a(x)'), (u_grid, 'Truth u(x)'), (u_gp, 'Recovered u(x)')]):
            ax = fig.add_subplot(2, 2, k + 1)
            cs = ax.contourf(XX, YY, Z, 50, cmap=plt.cm.coolwarm)
            ax.set_xlabel('x_1'); ax.set_ylabel('x_2'); ax.set_title(title)
            fig.colorbar(cs)
        fig.tight_layout()
        plt.show()


if __name__ == '__main__':
    main()

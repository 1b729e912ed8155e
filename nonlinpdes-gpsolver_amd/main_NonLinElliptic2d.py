#%%
"""Nonlinear elliptic equation -Delta u + alpha*u^m = f on [0,1]^2 with the GP solver on an MI355X.
Same command line as the reference's main_NonLinElliptic2d.py (README.md:15 of the reference):
    python main_NonLinElliptic2d.py --kernel Gaussian --kernel_parameter 0.2 --nugget 1e-13 --N_domain 900 --N_boundary 124 --GNsteps 4
The manufactured solution's right-hand side is written out analytically (the reference differentiates u with jax.grad)."""
import argparse

import numpy as onp

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, report_test_error, solve_forward, tensor_grid

UNIT_SQUARE = [[0, 1], [0, 1]]


def parse(argv=None):
    parser = argparse.ArgumentParser(description='NonLinElliptic equation GP solver')
    parser.add_argument("--alpha", type=float, default=1.0)
    parser.add_argument("--m", type=float, default=3.0)
    add_kernel_and_sampling(parser, 'Gaussian', 0.2, 1e-13, 900, 124)
    parser.add_argument("--pen_lambda", type=float, default=1e-10)      # for the relaxation approach
    add_gn_and_logs(parser, 'rdm', 4, method_choices=['elimination', 'relaxation'])
    return parser.parse_args(argv)


def manufactured(alpha, m):
    """u* = sin(pi x1) sin(pi x2) + 2 sin(4 pi x1) sin(4 pi x2) and f = -Laplace(u*) + alpha u*^m"""
    pi = onp.pi

    def u(x1, x2):
        return onp.sin(pi * x1) * onp.sin(pi * x2) + 2 * onp.sin(4 * pi * x1) * onp.sin(4 * pi * x2)

    def f(x1, x2):
        lap = -2 * pi ** 2 * onp.sin(pi * x1) * onp.sin(pi * x2) - 64 * pi ** 2 * onp.sin(4 * pi * x1) * onp.sin(4 * pi * x2)
        return -lap + alpha * (u(x1, x2) ** m)
    return u, f


def main(argv=None):
    cfg = parse(argv)
    u, f = manufactured(cfg.alpha, cfg.m)
    solver, show = solve_forward(cfg, "Nonlinear_elliptic", u, f, UNIT_SQUARE,
                                 solve_kwargs={'method': cfg.method, 'pen_lambda': cfg.pen_lambda}, verbose=cfg.print_hist)
    Xd = solver.eqn.X_domain
    solver.collocation_pts_err(u(Xd[:, 0], Xd[:, 1]))                    # error on the collocation points
    XX, YY, X_test = tensor_grid(60, *UNIT_SQUARE)                       # error on a 60 x 60 test grid
    report_test_error(solver, show, XX, YY, X_test, u(X_test[:, 0], X_test[:, 1]))


if __name__ == '__main__':
    main()

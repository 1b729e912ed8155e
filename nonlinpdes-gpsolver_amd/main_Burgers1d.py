#%%
"""Burgers equation u_t + alpha u u_x - nu u_xx = 0 on (t,x) in [0,1]x[-1,1] with the GP solver on an MI355X.
Same command line as the reference's main_Burgers1d.py (flags, defaults, --randomseed)."""
import argparse

import numpy as onp

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, report_test_error, seed_from, solve_forward, tensor_grid

SPACE_TIME = [[0, 1], [-1, 1]]


def parse(argv=None):
    parser = argparse.ArgumentParser(description='Burgers equation GP solver')
    parser.add_argument("--alpha", type=float, default=1.0)
    parser.add_argument("--nu", type=float, default=0.02)
    add_kernel_and_sampling(parser, 'anisotropic_Gaussian', [0.3, 0.05], 1e-5, 1000, 200, kp_nargs='+')
    add_gn_and_logs(parser, 'rdm', 8)
    parser.add_argument("--randomseed", type=int, default=0)
    return parser.parse_args(argv)


def initial_and_lateral(x1, x2):
    """-sin(pi x) at t = 0, zero on the lateral boundaries"""
    return -onp.sin(onp.pi * x2) * (x1 == 0) + 0 * (x2 == 0)


def cole_hopf_truth(nu, order=80):
    """exact solution by the Cole-Hopf transform, Gauss-Hermite quadrature with `order` nodes"""
    nodes, weights = onp.polynomial.hermite.hermgauss(order)

    def u_truth(t, x):
        y = x[..., None] - onp.sqrt(4 * nu * t[..., None]) * nodes
        e = weights * onp.exp(-onp.cos(onp.pi * y) / (2 * onp.pi * nu))
        return -onp.sum(onp.sin(onp.pi * y) * e, axis=-1) / onp.sum(e, axis=-1)
    return u_truth


def main(argv=None):
    cfg = parse(argv)
    seed_from(cfg)
    solver, show = solve_forward(cfg, "Burgers", initial_and_lateral, lambda x1, x2: 0, SPACE_TIME)
    XX, YY, X_test = tensor_grid(60, *SPACE_TIME)
    report_test_error(solver, show, XX, YY, X_test, cole_hopf_truth(cfg.nu)(X_test[:, 0], X_test[:, 1]))


if __name__ == '__main__':
    main()

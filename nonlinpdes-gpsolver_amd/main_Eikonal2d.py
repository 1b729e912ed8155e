#%%
"""Regularised Eikonal equation |grad u|^2 = f + eps*Delta u on [0,1]^2 with the GP solver on an MI355X.
Same command line as the reference's main_Eikonal2d.py."""
import argparse

from _driver_common import add_gn_and_logs, add_kernel_and_sampling, report_test_error, solve_forward, tensor_grid
from reference_solver.Cole_Hopf_for_Eikonal import solve_Eikonal

UNIT_SQUARE = [[0, 1], [0, 1]]


def parse(argv=None):
    parser = argparse.ArgumentParser(description='Eikonal equation GP solver')
    parser.add_argument("--eps", type=float, default=1e-1)
    add_kernel_and_sampling(parser, 'Gaussian', 0.2, 1e-5, 1000, 200)
    add_gn_and_logs(parser, 'zero', 8)
    return parser.parse_args(argv)


def main(argv=None):
    cfg = parse(argv)
    solver, show = solve_forward(cfg, "Eikonal", lambda x1, x2: 0, lambda x1, x2: 1, UNIT_SQUARE)
    # test points: interior of a 60 x 60 grid; truth by the Cole-Hopf transform + finite differences on the same grid
    n = 60
    _, _, X_test = tensor_grid(n, *UNIT_SQUARE, interior=True)
    XX, YY, truth = solve_Eikonal(n - 2, cfg.eps)
    report_test_error(solver, show, XX, YY, X_test, truth.flatten())


if __name__ == '__main__':
    main()

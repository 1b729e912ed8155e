"""Shared argparse helpers of the main_*.py drivers (same flag names, types and defaults as the reference's drivers)."""


def add_kernel_and_sampling(parser, kernel, kernel_parameter, nugget, N_domain, N_boundary, kp_nargs=None):
    parser.add_argument("--kernel", type=str, default=kernel)
    if kp_nargs:
        parser.add_argument("--kernel_parameter", type=float, nargs=kp_nargs, default=kernel_parameter)
    else:
        parser.add_argument("--kernel_parameter", type=float, default=kernel_parameter)
    parser.add_argument("--nugget", type=float, default=nugget)
    parser.add_argument("--nugget_type", type=str, default="adaptive", choices=["adaptive", "identity", 'none'])
    parser.add_argument("--sampled_type", type=str, default='random', choices=['random', 'grid'])
    parser.add_argument("--N_domain", type=int, default=N_domain)
    parser.add_argument("--N_boundary", type=int, default=N_boundary)


def add_gn_and_logs(parser, initial_sol, GNsteps, method_choices=None):
    if method_choices:
        parser.add_argument("--method", type=str, default='elimination', choices=method_choices)
    else:
        parser.add_argument("--method", type=str, default='elimination')
    parser.add_argument("--initial_sol", type=str, default=initial_sol)
    parser.add_argument("--GNsteps", type=int, default=GNsteps)
    parser.add_argument("--step_size", type=int, default=1)
    # type=bool as in the reference: any non-empty string is True; pass --show_figure "" to disable
    parser.add_argument("--print_hist", type=bool, default=True)
    parser.add_argument("--show_figure", type=bool, default=True)


def figures_enabled(cfg):
    """Figures need matplotlib and a display backend; without them the drivers run headless and say so."""
    if not cfg.show_figure:
        return False
    try:
        import matplotlib
        import os
        if not os.environ.get('DISPLAY') and matplotlib.get_backend().lower() not in ('agg',):
            matplotlib.use('Agg')
        return True
    except Exception as e:          # pragma: no cover
        print(f'[Figures] disabled ({e})')
        return False


def tensor_grid(n, range1, range2, interior=False):
    """n x n tensor grid on range1 x range2 (meshgrid order of the reference drivers); interior drops the boundary layer.
    Returns (XX, YY, points) with points of shape (n_eff^2, 2)."""
    import numpy as onp
    a = onp.linspace(range1[0], range1[1], n)
    b = onp.linspace(range2[0], range2[1], n)
    if interior:
        a, b = a[1:-1], b[1:-1]
    XX, YY = onp.meshgrid(a, b)
    return XX, YY, onp.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)


def seed_from(cfg):
    """Drivers with --randomseed seed numpy's legacy global generator first (collocation points, noise, initial guess)."""
    from numpy import random
    random.seed(cfg.randomseed)
    print(f"[Seeds] random seeds: {cfg.randomseed}")


def solve_forward(cfg, pde_type, bdy, rhs, domain, solve_kwargs=None, verbose=None):
    """The part every forward-problem driver shares: solver, equation, collocation points, Gauss-Newton solve (+ figures)."""
    import numpy as onp
    from src.solver import solver_GP
    show = figures_enabled(cfg)
    solver = solver_GP(cfg, PDE_type=pde_type)
    kw = {} if verbose is None else {'print_option': verbose}
    solver.set_equation(bdy=bdy, rhs=rhs, domain=onp.array(domain), **kw)
    solver.auto_sample(cfg.N_domain, cfg.N_boundary, sampled_type=cfg.sampled_type, **kw)
    if show:
        solver.show_sample()
    solver.solve(**dict(solve_kwargs or {}, **kw))
    if show:
        solver.show_loss_hist()
    return solver, show


def report_test_error(solver, show, XX, YY, X_test, truth):
    solver.test(X_test)
    solver.get_test_error(truth)
    if show:
        solver.contour_of_test_err(XX, YY)

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

"""solver_GP facade with the reference's interface (src/solver.py:41-206): set_equation -> auto_sample -> solve -> test
-> errors, same bracket-tagged log lines.  Plot helpers use matplotlib when it is importable (no LaTeX requirement)."""
import numpy as onp

from ._runtime import get_context
from .InverseProblems import Darcy_flow2d
from .PDEs import Burgers, Eikonal, Nonlinear_elliptic2d

# PDE_type -> (factory, header lines printed by set_equation)
_EQUATIONS = {
    'Nonlinear_elliptic': (
        lambda c, **k: Nonlinear_elliptic2d(alpha=c.alpha, m=c.m, **k),
        lambda c: ['[Equation type] Nonlinear elliptic equation', '[Equation form] - \\Delta u + alpha*u^m = f'],
        lambda c: f'[Equation parameter] alpha = {c.alpha}, m = {c.m}'),
    'Burgers': (
        lambda c, **k: Burgers(alpha=c.alpha, nu=c.nu, **k),
        lambda c: ['[Equation type] Burgers equation', '[Equation form] u_t+ alpha u u_x- nu u_xx=0'],
        lambda c: f'[Equation parameter] alpha = {c.alpha}, m = {c.nu}'),        # sic: the reference prints "m ="
    'Eikonal': (
        lambda c, **k: Eikonal(eps=c.eps, **k),
        lambda c: ['[Equation type] Eikonal equation', '[Equation form] |grad u|^2 = f + eps*Delta u'],
        lambda c: f'[Equation parameter] eps = {c.eps}'),
    'Darcy_flow2d': (
        lambda c, **k: Darcy_flow2d(**k),
        lambda c: ['[Inverse problem type] Darcy flow 2d', '[Inverse problem form] -div(a grad u) = f, infer a from f and some observed u'],
        None),
}


# every bracket-tagged log line of the facade, keyed by event (texts are the reference's, src/solver.py:41-206)
_LOG = {
    'start': '\n Solver started',
    'domain': '[Equation domain] [{d[0][0]},{d[0][1]}]*[{d[1][0]},{d[1][1]}]',
    'data': '[Equation data] Right hand side and boundary values set by the user',
    'pts_user': '[Sample points] Collocation points sampled, specified by the user',
    'pts_auto': '[Sample points] Collocation points sampled, type {kind}',
    'pts_n': '[Sample points] N_domain = {e.N_domain}, N_boundary = {e.N_boundary}',
    'pts_n_ip': '[Sample points] N_domain = {e.N_domain}, N_boundary = {e.N_boundary}, N_data = {e.N_data}',
    'obs': '[Observed Data] Get observed data from solving the PDE using FD and interpolation',
    'obs_noise': '[Observed Data] Noise level {noise}',
    'kernel': '[Kernel] {c.kernel}',
    'kernel_par': '[Kernel parameter]: {c.kernel_parameter}',
    'gram': '[Gram matrix] Finish assembly of the Gram matrix, nugget {c.nugget}, type {c.nugget_type}',
    'chol': '[Gram matrix] Finish Cholesky factorization of the Gram matrix',
    'gn_start': '[Gauss Newton] Start Gauss Newton iteration',
    'gn_method': '[Gauss Newton] {method} approaches',
    'gn_done': '[Gauss Newton] Gauss Newton iteration finished',
    'pts_err': '[Calculating collocation errors...]',
    'pts_max': '[Collocation point error] Max error {v}',
    'pts_l2': '[Collocation point error] L2 error {v}',
    'testing': '[Testing...] Number of test points: {n}',
    'test_max': '[Test error] Max error {v}',
    'test_l2': '[Test error] L2 error {v}',
}


def _say(enabled, *keys, **fmt):
    if enabled:
        for k in keys:
            print(_LOG[k].format(**fmt))


def _as_vector(truth, n):
    """the user's truth as n float64 values (a scalar truth -- e.g. 0 -- is broadcast like under jnp)"""
    t = onp.asarray(truth, dtype=onp.float64)
    return onp.full(n, float(t)) if t.ndim == 0 else t.ravel()


class solver_GP(object):
    """Same public surface as the reference's solver_GP; the work happens in the equation objects (src/PDEs.py,
    src/InverseProblems.py), which drive libgpk."""

    def __init__(self, cfg=None, PDE_type="Nonlinear_elliptic"):
        self.config = cfg
        self.PDE_type = PDE_type

    def set_equation(self, bdy=None, rhs=None, domain=onp.array([[0, 1], [0, 1]]), print_option=True):
        if self.PDE_type not in _EQUATIONS:
            return
        make, header, params = _EQUATIONS[self.PDE_type]
        self.eqn = make(self.config, bdy=bdy, rhs=rhs, domain=domain)
        if print_option:
            print(_LOG['start'])
            for line in header(self.config):
                print(line)
            _say(True, 'domain', d=domain)
            if params is not None:
                print(params(self.config))
            _say(True, 'data')

    # ---- collocation (and data) points: given by the caller or drawn by the equation object's sampler -------------------
    def get_sample(self, X_domain, X_boundary, print_option=True):
        # (the reference passes `self` twice here and raises TypeError; this one works)
        self.eqn.get_sampled_points(X_domain, X_boundary)
        _say(print_option, 'pts_user', 'pts_n', e=self.eqn)

    def auto_sample(self, N_domain, N_boundary, sampled_type='random', print_option=True):
        self.eqn.sampled_pts(N_domain, N_boundary, sampled_type=sampled_type)
        _say(print_option, 'pts_auto', 'pts_n', e=self.eqn, kind=sampled_type)

    def get_sample_IP(self, X_domain, X_boundary, X_data, print_option=True):
        self.eqn.get_sampled_points(X_domain, X_boundary, X_data)
        _say(print_option, 'pts_user', 'pts_n_ip', e=self.eqn)

    def auto_sample_IP(self, N_domain, N_boundary, N_data, sampled_type='random', print_option=True):
        self.eqn.sampled_pts(N_domain, N_boundary, N_data, sampled_type=sampled_type)
        _say(print_option, 'pts_auto', 'pts_n_ip', e=self.eqn, kind=sampled_type)

    def get_observed_data(self, data_u, noise_level, print_option=True):
        self.eqn.get_observation(data_u, noise_level)
        _say(print_option, 'obs', 'obs_noise', noise=noise_level)

    # ---- Gram matrix -> Cholesky -> Gauss-Newton -------------------------------------------------------------------------
    def solve(self, method='elimination', pen_lambda=1e-10, print_option=True):
        c, eqn = self.config, self.eqn
        _say(print_option, 'kernel', 'kernel_par', c=c)
        eqn.Gram_matrix(kernel=c.kernel, kernel_parameter=c.kernel_parameter, nugget=c.nugget, nugget_type=c.nugget_type)
        _say(print_option, 'gram', c=c)
        eqn.Gram_Cholesky()
        _say(print_option, 'chol', 'gn_start', 'gn_method', method=method)
        gn = dict(max_iter=c.GNsteps, step_size=c.step_size, initial_sol=c.initial_sol, print_hist=c.print_hist)
        if method == 'elimination':
            eqn.GN_method(**gn)
        elif method == 'relaxation':
            eqn.GN_relaxed_method(pen_lambda=pen_lambda, **gn)
        _say(print_option, 'gn_done')

    # ---- errors (definitions of src/solver.py:175,191: root of the sum of squares over N_domain / N_test), reduced on the device
    # (gpk_error_metrics: |truth - value| per point, its maximum and sqrt(sum of squares / n) in one pass) ------------------------------
    def collocation_pts_err(self, truth, print_option=True):
        _say(print_option, 'pts_err')
        self.pts_err_all, self.pts_max_err, self.pts_L2_err = get_context().error_metrics(_as_vector(truth, self.eqn.N_domain),
                                                                                             self.eqn.sol_sampled_pts)
        _say(print_option, 'pts_max', v=self.pts_max_err)
        _say(print_option, 'pts_l2', v=self.pts_L2_err)

    def test(self, X_test, print_option=True):
        _say(print_option, 'testing', n=X_test.shape[0])
        self.eqn.extend_sol(X_test)

    def get_test_error(self, truth, print_option=True):
        self.truth = truth
        self.test_err_all, self.test_max_err, self.test_L2_err = get_context().error_metrics(_as_vector(truth, self.eqn.N_test),
                                                                                               self.eqn.extended_sol)
        _say(print_option, 'test_max', v=self.test_max_err)
        _say(print_option, 'test_l2', v=self.test_L2_err)

    # ---- figures (cosmetic; need matplotlib only): src/_figures.py ---------------------------------------------------------
    def show_sample(self):
        from . import _figures
        _figures.scatter_points(self.eqn, False, 'Collocation points')

    def show_sample_IP(self):
        from . import _figures
        _figures.scatter_points(self.eqn, True, 'Collocation and data points')

    def show_loss_hist(self):
        from . import _figures
        _figures.loss_history(self.eqn)

    def contour_of_test_err(self, XX, YY):
        from . import _figures
        self.XX, self.YY = XX, YY
        _figures.error_contour(XX, YY, self.test_err_all)

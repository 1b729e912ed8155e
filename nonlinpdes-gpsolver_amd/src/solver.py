"""solver_GP facade with the reference's interface (src/solver.py:41-206): set_equation -> auto_sample -> solve -> test
-> errors, same bracket-tagged log lines.  Plot helpers use matplotlib when it is importable (no LaTeX requirement)."""
import numpy as onp

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


def _plt():
    import matplotlib
    import matplotlib.pyplot as plt
    return plt


class solver_GP(object):
    def __init__(self, cfg=None, PDE_type="Nonlinear_elliptic"):
        self.config = cfg
        self.PDE_type = PDE_type

    def set_equation(self, bdy=None, rhs=None, domain=onp.array([[0, 1], [0, 1]]), print_option=True):
        if self.PDE_type not in _EQUATIONS:
            return
        make, header, params = _EQUATIONS[self.PDE_type]
        self.eqn = make(self.config, bdy=bdy, rhs=rhs, domain=domain)
        if print_option:
            print('\n Solver started')
            for line in header(self.config):
                print(line)
            print(f'[Equation domain] [{domain[0,0]},{domain[0,1]}]*[{domain[1,0]},{domain[1,1]}]')
            if params is not None:
                print(params(self.config))
            print('[Equation data] Right hand side and boundary values set by the user')

    # ---- sampling ------------------------------------------------------------------------------------------------------
    def get_sample(self, X_domain, X_boundary, print_option=True):
        # (the reference passes `self` twice here and raises TypeError; this one works)
        self.eqn.get_sampled_points(X_domain, X_boundary)
        if print_option:
            print('[Sample points] Collocation points sampled, specified by the user')
            print(f'[Sample points] N_domain = {self.eqn.N_domain}, N_boundary = {self.eqn.N_boundary}')

    def auto_sample(self, N_domain, N_boundary, sampled_type='random', print_option=True):
        self.eqn.sampled_pts(N_domain, N_boundary, sampled_type=sampled_type)
        if print_option:
            print(f'[Sample points] Collocation points sampled, type {sampled_type}')
            print(f'[Sample points] N_domain = {self.eqn.N_domain}, N_boundary = {self.eqn.N_boundary}')

    def get_sample_IP(self, X_domain, X_boundary, X_data, print_option=True):
        self.eqn.get_sampled_points(X_domain, X_boundary, X_data)
        if print_option:
            print('[Sample points] Collocation points sampled, specified by the user')
            print(f'[Sample points] N_domain = {self.eqn.N_domain}, N_boundary = {self.eqn.N_boundary}, N_data = {self.eqn.N_data}')

    def auto_sample_IP(self, N_domain, N_boundary, N_data, sampled_type='random', print_option=True):
        self.eqn.sampled_pts(N_domain, N_boundary, N_data, sampled_type=sampled_type)
        if print_option:
            print(f'[Sample points] Collocation points sampled, type {sampled_type}')
            print(f'[Sample points] N_domain = {self.eqn.N_domain}, N_boundary = {self.eqn.N_boundary}, N_data = {self.eqn.N_data}')

    def get_observed_data(self, data_u, noise_level, print_option=True):
        self.eqn.get_observation(data_u, noise_level)
        if print_option:
            print('[Observed Data] Get observed data from solving the PDE using FD and interpolation')
            print(f'[Observed Data] Noise level {noise_level}')

    # ---- solve ---------------------------------------------------------------------------------------------------------
    def solve(self, method='elimination', pen_lambda=1e-10, print_option=True):
        c = self.config
        if print_option:
            print('[Kernel] ' + c.kernel)
            print(f'[Kernel parameter]: {c.kernel_parameter}')
        self.eqn.Gram_matrix(kernel=c.kernel, kernel_parameter=c.kernel_parameter, nugget=c.nugget, nugget_type=c.nugget_type)
        if print_option:
            print(f'[Gram matrix] Finish assembly of the Gram matrix, nugget {c.nugget}, type {c.nugget_type}')
        self.eqn.Gram_Cholesky()
        if print_option:
            print('[Gram matrix] Finish Cholesky factorization of the Gram matrix')
            print('[Gauss Newton] Start Gauss Newton iteration')
            print(f'[Gauss Newton] {method} approaches')
        if method == 'elimination':
            self.eqn.GN_method(max_iter=c.GNsteps, step_size=c.step_size, initial_sol=c.initial_sol, print_hist=c.print_hist)
        elif method == 'relaxation':
            self.eqn.GN_relaxed_method(max_iter=c.GNsteps, step_size=c.step_size, initial_sol=c.initial_sol,
                                       pen_lambda=pen_lambda, print_hist=c.print_hist)
        if print_option:
            print('[Gauss Newton] Gauss Newton iteration finished')

    # ---- errors --------------------------------------------------------------------------------------------------------
    def collocation_pts_err(self, truth, print_option=True):
        if print_option:
            print('[Calculating collocation errors...]')
        self.pts_err_all = abs(onp.asarray(truth) - self.eqn.sol_sampled_pts)
        self.pts_max_err = onp.max(self.pts_err_all)
        self.pts_L2_err = onp.sqrt(onp.sum(self.pts_err_all ** 2) / (self.eqn.N_domain))
        if print_option:
            print(f'[Collocation point error] Max error {self.pts_max_err}')
            print(f'[Collocation point error] L2 error {self.pts_L2_err}')

    def test(self, X_test, print_option=True):
        if print_option:
            print(f'[Testing...] Number of test points: {X_test.shape[0]}')
        self.eqn.extend_sol(X_test)

    def get_test_error(self, truth, print_option=True):
        self.truth = truth
        self.test_err_all = abs(onp.asarray(truth) - self.eqn.extended_sol)
        self.test_max_err = onp.max(self.test_err_all)
        self.test_L2_err = onp.sqrt(onp.sum(self.test_err_all ** 2) / (self.eqn.N_test))
        if print_option:
            print(f'[Test error] Max error {self.test_max_err}')
            print(f'[Test error] L2 error {self.test_L2_err}')

    # ---- figures (cosmetic; need matplotlib only) ----------------------------------------------------------------------
    def _scatter(self, with_data, title):
        plt = _plt()
        fig = plt.figure()
        ax = fig.add_subplot(111)
        e = self.eqn
        series = [(e.X_domain, 'Interior nodes'), (e.X_boundary, 'Boundary nodes')]
        if with_data:
            series.append((e.X_domain[:e.N_data], 'Data nodes'))
        for X, label in series:
            ax.scatter(X[:, 0], X[:, 1], marker=None if with_data else 'x', label=label).set_clip_on(False)
        ax.legend(loc="upper right")
        plt.title(title)

    def show_sample(self):
        self._scatter(False, 'Collocation points')

    def show_sample_IP(self):
        self._scatter(True, 'Collocation and data points')

    def show_loss_hist(self):
        plt = _plt()
        plt.figure()
        plt.plot(onp.arange(self.eqn.max_iter + 1), self.eqn.loss_hist)
        plt.yscale("log")
        plt.title('Loss function history')
        plt.xlabel('Gauss-Newton step')

    def contour_of_test_err(self, XX, YY):
        plt = _plt()
        fig = plt.figure()
        ax = fig.add_subplot(111)
        cs = ax.contourf(XX, YY, self.test_err_all.reshape(XX.shape), 50, cmap=plt.cm.coolwarm)
        self.XX = XX
        self.YY = YY
        plt.xlabel('x_1')
        plt.ylabel('x_2')
        plt.title('Contour of errors')
        fig.colorbar(cs)
        plt.show()

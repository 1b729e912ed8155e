"""Equation classes with the reference's public API (src/PDEs.py): Nonlinear_elliptic2d, Burgers, Eikonal.

Same constructor arguments, methods and attributes; no JAX.  The Gram matrix, its Cholesky factor and the Gauss-Newton
iterate live in GPU memory (libgpk.so); `Theta`, `L`, ... are materialised as numpy arrays only when read.
Differences that are deliberate: no stale jit cache when Gram_matrix is called twice; user callbacks
may be numpy-vectorised or plain scalar Python functions.
"""
import os

import numpy as onp
from numpy import random

import gpk

from ._runtime import eval_callback, get_context
from .sample_points import sampled_pts_grid, sampled_pts_rdm

_NAN_MSG = '[Error] Loss is nan: maybe nugget is too small!'


class _GPEquation(object):
    """Machinery shared by the three PDE classes (and by InverseProblems.Darcy_flow2d)."""
    _layout = None              # Gram layout name
    _system = None              # Gauss-Newton system name
    _time_dependent = False
    _blocks_per_point = 1       # unknown groups per collocation point

    # ---- user data -------------------------------------------------------------------------------------------------
    def get_bd(self, x1, x2):
        return self.bdy(x1, x2)

    def get_rhs(self, x1, x2):
        return self.rhs(x1, x2)

    def _set_points(self, X_domain, X_boundary):
        self.X_domain = onp.asarray(X_domain, dtype=onp.float64)
        self.N_domain = self.X_domain.shape[0]
        self.X_boundary = onp.asarray(X_boundary, dtype=onp.float64)
        self.N_boundary = self.X_boundary.shape[0]
        self.rhs_f = eval_callback(self.get_rhs, self.X_domain[:, 0], self.X_domain[:, 1])
        self.bdy_g = eval_callback(self.get_bd, self.X_boundary[:, 0], self.X_boundary[:, 1])
        self._drop_device_state()

    def sampled_pts(self, N_domain, N_boundary, sampled_type='random'):
        if sampled_type == 'random':
            X_domain, X_boundary = sampled_pts_rdm(N_domain, N_boundary, self.domain, time_dependent=self._time_dependent)
        elif sampled_type == 'grid':
            X_domain, X_boundary = sampled_pts_grid(N_domain, N_boundary, self.domain, time_dependent=self._time_dependent)
        else:
            raise UnboundLocalError("local variable 'X_domain' referenced before assignment")
        self._set_points(X_domain, X_boundary)

    def get_sampled_points(self, X_domain, X_boundary):
        self._set_points(X_domain, X_boundary)

    # ---- device state ----------------------------------------------------------------------------------------------
    def _drop_device_state(self):
        for name in ('_dTheta', '_dL'):
            a = self.__dict__.pop(name, None)
            if a is not None:
                a.free()
        p = self.__dict__.pop('_prob', None)
        if p is not None:
            p.free()                                           # (workspace, inverted blocks, prepared operators: everything but the factors)
        self.__dict__.pop('_Theta_host', None)
        self.__dict__.pop('_L_host', None)

    def _n_unknowns(self):
        return self._blocks_per_point * self.N_domain

    def _gn_params(self):
        raise NotImplementedError

    def _problem(self):
        if getattr(self, '_prob', None) is None:
            if getattr(self, '_dL', None) is None:
                raise RuntimeError('call Gram_matrix() and Gram_Cholesky() first')
            p0, p1, lam = self._gn_params()
            # GPK_STRUCTURED=1 / 2 (opt-in, elliptic system only; 2 adds the Gram level, gpk_gn_gram_prepare): the z-independent solves are done once and every step forms
            # [L^{-1}A(z) | L^{-1}F(z)] from them (gpk_gn_structured_prepare) -- same iterates, about half the time per step
            # (round 6: GPK_STRUCTURED=1 also for the Burgers and Eikonal systems, and the Darcy system in InverseProblems.py: A(z) = A1 diag(d(z)) + A2)
            structured = int(os.environ.get('GPK_STRUCTURED', '0') or 0)
            # (2 = the Gram level: since round 6 for every system with a structured form)
            self._prob = gpk.GNProblem(get_context(), self._system, self.N_domain, self.N_boundary, self.rhs_f, self.bdy_g,
                                       self._dL, p0=p0, p1=p1, pen_lambda=lam, structured=structured)
        return self._prob

    # ---- Gram matrix + Cholesky ------------------------------------------------------------------------------------
    def _assemble(self, kernel, kernel_parameter, nugget, nugget_type):
        if nugget_type not in ('adaptive', 'identity', 'none'):
            raise AttributeError(f"nugget_type {nugget_type!r}: the reference leaves self.Theta unset here")
        ctx = get_context()
        self._drop_device_state()
        self.nugget_type = nugget_type
        self.nugget = nugget
        self.kernel = kernel
        self.kernel_parameter = kernel_parameter
        self._dTheta, ratios = ctx.assemble(self._layout, kernel, kernel_parameter, self.X_domain, self.X_boundary,
                                            nugget, nugget_type)
        return ratios

    @property
    def Theta(self):
        """nugget-regularised Gram matrix as a numpy array (downloaded on first access)"""
        if '_Theta_host' not in self.__dict__:
            src = self.__dict__.get('_dTheta')
            if src is None:
                raise AttributeError('Theta: call Gram_matrix() first')
            self._Theta_host = src.download()
        return self._Theta_host

    def Gram_Cholesky(self):
        ctx = get_context()
        if getattr(self, '_dTheta', None) is None:
            raise AttributeError("Theta: call Gram_matrix() first")
        old = self.__dict__.pop('_dL', None)
        if old is not None:
            old.free()
        self.__dict__.pop('_L_host', None)
        self._dL = self._dTheta.clone()                   # Theta stays readable, as in the reference (HBM is plentiful)
        self.chol_info = ctx.potrf(self._dL)              # >0: first non-positive pivot; NaNs propagate like JAX
        if self.chol_info != 0:
            # the reference's try/except never fires under JAX: it carries on with NaNs, and so do we
            print('[Warning] Cholesky factorization met a non-positive pivot at index', self.chol_info,
                  '(the reference would silently continue with NaNs): maybe nugget is too small!')

    @property
    def L(self):
        if '_L_host' not in self.__dict__:
            if getattr(self, '_dL', None) is None:
                raise AttributeError('L: call Gram_Cholesky() first')
            self._L_host = onp.tril(self._dL.download())
        return self._L_host

    # ---- loss / gradient / Hessian -----------------------------------------------------------------------------------
    def _z(self, z):
        return get_context().array(onp.asarray(z, dtype=onp.float64).ravel())

    def loss(self, z):
        return get_context().gn_loss(self._problem(), self._z(z))

    def grad_loss(self, z):
        _, g = get_context().gn_hessian_grad(self._problem(), self._z(z))
        return g

    def _hessian(self, z):
        H, _ = get_context().gn_hessian_grad(self._problem(), self._z(z))
        return H

    def _measurement(self, z):
        return get_context().gn_measurement(self._problem(), self._z(z))

    def _tri_loss(self, vec):
        """||L^{-1} vec||^2 for a host vector (used by GN_loss)."""
        ctx = get_context()
        d = ctx.array(vec)
        ctx.trsm(self._dL, d, trans=False, nrhs=1)
        w = d.download()
        return float(w @ w)

    # ---- Gauss-Newton ----------------------------------------------------------------------------------------------
    def _initial(self, initial_sol, n):
        if initial_sol == 'rdm':
            return random.normal(0.0, 1.0, (n))
        raise UnboundLocalError("local variable 'sol' referenced before assignment")   # as the reference

    def _check_step_info(self, it, info):
        """info > 0: the Cholesky of the Gauss-Newton matrix H of step `it` met a non-positive pivot (LAPACK numbering).  The reference
        solves H with jnp.linalg.solve (src/PDEs.py:118), which never reports anything and carries on with whatever comes out; here the step
        has been taken with NaNs from that pivot on, exactly as visible in the next loss -- say so at once, in the reference's wording."""
        self.step_info = getattr(self, 'step_info', [])
        self.step_info.append(int(info))
        if info > 0:
            print('[Error] Cholesky factorization of the Gauss-Newton matrix failed at step', it, '(non-positive pivot at index', info,
                  '): maybe nugget is too small!')

    def _gn_iterate(self, prob, sol, max_iter, step_size, print_hist, check_nan=True):
        self.step_info = []
        ctx = get_context()
        z = ctx.array(sol)
        loss_hist = []

        def record(it, value):
            loss_hist.append(value)
            if check_nan and onp.isnan(value):
                print(_NAN_MSG)
            if print_hist:
                if it == 0:
                    print('iter = 0', 'Loss =', value)
                else:
                    print('iter = ', it, 'Gauss-Newton step size =', step_size, ' Loss = ', value)
        # The loss history is the reference's: J(z_0), then J(z_k) after every update (src/PDEs.py:108-124).  gpk_gn_step returns the
        # loss of the iterate it STARTS from -- since round 5 by true substitution with the factor (one vector; its chain runs on the handle's side
        # stream beside the end of the step: exact to rounding, like gpk_gn_loss; round 6: in the structured modes too) -- so max_iter steps yield
        # J(z_0) .. J(z_{max_iter-1}) and one gpk_gn_loss closes the history.  Rounds 2-4 took that number from the F column of the
        # GEMM-only solve (~1e-8 relative error at nugget <= 1e-12 near convergence: gpk_tune(52, 0)) and round 4 therefore called
        # gpk_gn_loss after every step; GPK_SEPARATE_LOSS=1 keeps that sequence (same numbers to rounding, one more solve per step).
        if os.environ.get('GPK_SEPARATE_LOSS', '0') != '1':
            for it in range(max_iter):
                loss_in, info = ctx.gn_step(prob, z, step_size)   # loss of the iterate the step starts from
                record(it, loss_in)
                self._check_step_info(it + 1, info)
            record(max_iter, ctx.gn_loss(prob, z))
        else:
            record(0, ctx.gn_loss(prob, z))
            for it in range(1, max_iter + 1):
                _, info = ctx.gn_step(prob, z, step_size)
                self._check_step_info(it, info)
                record(it, ctx.gn_loss(prob, z))
        self.max_iter = max_iter
        self.step_size = step_size
        self.loss_hist = loss_hist
        return z.download()

    def extend_sol(self, X_test):
        ctx = get_context()
        X_test = onp.asarray(X_test, dtype=onp.float64)
        coeff = ctx.array(self.sol_vec)
        ctx.potrs(self._dL, coeff, nrhs=1)                         # L^{-T} L^{-1} sol_vec
        self.X_test = X_test
        self.N_test = X_test.shape[0]
        self.extended_sol = ctx.extend(self._layout, self.kernel, self.kernel_parameter, X_test, self.X_domain,
                                       self.X_boundary, coeff).download()


class Nonlinear_elliptic2d(_GPEquation):
    """-Delta u + alpha*u^m = f on a rectangle (reference src/PDEs.py:18-208)."""
    _layout = 'Nonlinear_elliptic'
    _system = 'Nonlinear_elliptic'

    def __init__(self, alpha=1.0, m=3, bdy=None, rhs=None, domain=onp.array([[0, 1], [0, 1]])):
        self.alpha = alpha
        self.m = m
        self.bdy = bdy
        self.rhs = rhs
        self.domain = domain

    def _gn_params(self):
        return float(self.alpha), float(self.m), 0.0

    def Gram_matrix(self, kernel='Gaussian', kernel_parameter=0.2, nugget=1e-8, nugget_type='adaptive'):
        ratios = self._assemble(kernel, kernel_parameter, nugget, nugget_type)
        if nugget_type == 'adaptive':
            self.ratio = ratios[0]

    def GN_loss(self, z, z_old):
        z = onp.asarray(z, float); z_old = onp.asarray(z_old, float)
        zz = onp.concatenate([self.alpha * self.m * (z_old ** (self.m - 1)) * (z - z_old), z, self.bdy_g])
        return self._tri_loss(zz)

    def Hessian_GN(self, z, z_old):
        return self._hessian(z_old)          # hessian(GN_loss)(z, z_old) does not depend on z (quadratic in z)

    def GN_method(self, max_iter=3, step_size=1, initial_sol='rdm', print_hist=True):
        sol = self._initial(initial_sol, self.N_domain)
        self.init_sol = sol
        sol = self._gn_iterate(self._problem(), sol, max_iter, step_size, print_hist)
        self.sol_vec = onp.concatenate([self.alpha * (sol ** self.m) - self.rhs_f, sol, self.bdy_g])
        self.sol_sampled_pts = sol

    # ---- relaxed (penalised) formulation, reference src/PDEs.py:137-201 ----
    def _relaxed_problem(self, pen_lambda):
        key = ('_prob_relaxed', float(pen_lambda))
        if getattr(self, '_prob_relaxed_key', None) != key:
            self._prob_relaxed = gpk.GNProblem(get_context(), 'Nonlinear_elliptic_relaxed', self.N_domain, self.N_boundary,
                                               self.rhs_f, self.bdy_g, self._dL, p0=float(self.alpha), p1=float(self.m),
                                               pen_lambda=float(pen_lambda))
            self._prob_relaxed_key = key
        return self._prob_relaxed

    def loss_relaxed(self, z, pen_lambda):
        return get_context().gn_loss(self._relaxed_problem(pen_lambda), self._z(z))

    def grad_loss_relaxed(self, z, pen_lambda):
        return get_context().gn_hessian_grad(self._relaxed_problem(pen_lambda), self._z(z))[1]

    def GN_loss_relaxed(self, z, z_old, pen_lambda):
        z = onp.asarray(z, float); z_old = onp.asarray(z_old, float)
        Nd = self.N_domain
        v, w, w_old = z[:Nd], z[Nd:], z_old[Nd:]
        ss2 = -v + self.alpha * self.m * (w_old ** (self.m - 1)) * (w - w_old) - self.rhs_f
        return self._tri_loss(onp.concatenate([v, w, self.bdy_g])) + float(ss2 @ ss2) / pen_lambda

    def Hessian_GN_relaxed(self, z, z_old, pen_lambda):
        return get_context().gn_hessian_grad(self._relaxed_problem(pen_lambda), self._z(z_old))[0]

    def GN_relaxed_method(self, max_iter=3, step_size=1, initial_sol='rdm', pen_lambda=1e-10, print_hist=True):
        print(f'Relaxed approach: penalization parameter = {pen_lambda}')
        sol = self._initial(initial_sol, 2 * self.N_domain)
        self.init_sol = sol
        sol = self._gn_iterate(self._relaxed_problem(pen_lambda), sol, max_iter, step_size, print_hist)
        self.sol_vec = onp.concatenate([sol, self.bdy_g])
        self.sol_sampled_pts = sol[self.N_domain:]


class Burgers(_GPEquation):
    """u_t + alpha u u_x - nu u_xx = 0 on (t,x) in a rectangle (reference src/PDEs.py:211-350)."""
    _layout = 'Burgers'
    _system = 'Burgers'
    _time_dependent = True
    _blocks_per_point = 3

    def __init__(self, alpha=1.0, nu=0.2, bdy=None, rhs=None, domain=onp.array([[0, 1], [-1, 1]])):
        self.alpha = alpha
        self.nu = nu
        self.bdy = bdy
        self.rhs = rhs
        self.domain = domain

    def _gn_params(self):
        return float(self.alpha), float(self.nu), 0.0

    def Gram_matrix(self, kernel='anisotropic_Gaussian', kernel_parameter=[1 / 3, 1 / 20], nugget=1e-5, nugget_type='adaptive'):
        ratios = self._assemble(kernel, kernel_parameter, nugget, nugget_type)
        if nugget_type == 'adaptive':
            self.ratio = list(ratios[:3])

    def Hessian_GN(self, z):                 # ONE argument in the reference (src/PDEs.py:295)
        return self._hessian(z)

    def GN_method(self, max_iter=10, step_size=1, initial_sol='rdm', print_hist=True):
        sol = self._initial(initial_sol, 3 * self.N_domain)
        self.init_sol = sol
        sol = self._gn_iterate(self._problem(), sol, max_iter, step_size, print_hist)
        Nd = self.N_domain
        v0, v2, v3 = sol[:Nd], sol[Nd:2 * Nd], sol[2 * Nd:]
        self.sol_vec = onp.concatenate((self.nu * v3 + self.rhs_f - self.alpha * v0 * v2, v2, v3, v0, self.bdy_g), axis=0)
        self.sol_sampled_pts = v0


class Eikonal(_GPEquation):
    """|grad u|^2 = f^2 + eps*Delta u (reference src/PDEs.py:352-505)."""
    _layout = 'Eikonal'
    _system = 'Eikonal'
    _blocks_per_point = 3

    def __init__(self, eps=3, bdy=None, rhs=None, domain=onp.array([[0, 1], [0, 1]])):
        self.eps = eps
        self.bdy = bdy
        self.rhs = rhs
        self.domain = domain

    def _gn_params(self):
        return float(self.eps), 0.0, 0.0

    def Gram_matrix(self, kernel='Gaussian', kernel_parameter=0.2, nugget=1e-8, nugget_type='adaptive'):
        self._assemble(kernel, kernel_parameter, nugget, nugget_type)     # the reference does not store `ratio` here

    def GN_loss(self, z, z_old):
        z = onp.asarray(z, float); z_old = onp.asarray(z_old, float)
        Nd = self.N_domain
        v1o, v2o = z_old[Nd:2 * Nd], z_old[2 * Nd:]
        v0, v1, v2 = z[:Nd], z[Nd:2 * Nd], z[2 * Nd:]
        v3 = -(self.rhs_f ** 2 - 2 * v1 * v1o - 2 * v2 * v2o) / self.eps
        return self._tri_loss(onp.concatenate([v1, v2, v3, v0, self.bdy_g]))

    def Hessian_GN(self, z, z_old):
        return self._hessian(z_old)

    def _initial(self, initial_sol, n):
        if initial_sol == 'zero':
            return onp.zeros(n)
        return super()._initial(initial_sol, n)

    def GN_method(self, max_iter=3, step_size=1, initial_sol='rdm', print_hist=True):
        sol = self._initial(initial_sol, 3 * self.N_domain)
        self.init_sol = sol
        sol = self._gn_iterate(self._problem(), sol, max_iter, step_size, print_hist)
        Nd = self.N_domain
        v0, v1, v2 = sol[:Nd], sol[Nd:2 * Nd], sol[2 * Nd:]
        v3 = -(self.rhs_f ** 2 - v1 ** 2 - v2 ** 2) / self.eps
        self.sol_vec = onp.concatenate([v1, v2, v3, v0, self.bdy_g])
        self.sol_sampled_pts = v0

"""Darcy_flow2d inverse problem with the reference's public API (src/InverseProblems.py:16-196): two coupled GPs
(log-permeability a, pressure u), two Gram matrices / Cholesky factors, 6*N_domain unknowns, noisy observations of u
at the first N_data collocation points."""
import os

import numpy as onp
from numpy import random

import gpk

from ._runtime import eval_callback, get_context
from .PDEs import _GPEquation, _NAN_MSG  # noqa: F401
from .sample_points import sampled_pts_grid, sampled_pts_rdm


class Darcy_flow2d(_GPEquation):
    _system = 'Darcy_flow2d'
    _blocks_per_point = 6

    def __init__(self, bdy=None, rhs=None, domain=onp.array([[0, 1], [0, 1]])):
        self.bdy = bdy
        self.rhs = rhs
        self.domain = domain

    # ---- points: the first N_data collocation points carry the observations --------------------------------------
    def sampled_pts(self, N_domain, N_boundary, N_data, sampled_type='random'):
        if sampled_type == 'random':
            X_domain, X_boundary = sampled_pts_rdm(N_domain, N_boundary, self.domain, time_dependent=False)
        elif sampled_type == 'grid':
            X_domain, X_boundary = sampled_pts_grid(N_domain, N_boundary, self.domain, time_dependent=False)
        else:
            raise UnboundLocalError("local variable 'X_domain' referenced before assignment")
        self._set_points(X_domain, X_boundary)
        self.X_data = self.X_domain[0:N_data, :]
        self.N_data = N_data

    def get_sampled_points(self, X_domain, X_boundary, X_data):
        self._set_points(X_domain, X_boundary)
        self.X_data = onp.asarray(X_data, dtype=onp.float64)
        self.N_data = self.X_data.shape[0]

    def get_observation(self, data_u, noise_level):
        data_u = onp.asarray(data_u, dtype=onp.float64)
        self.data_u = data_u + noise_level * random.normal(0, 1.0, onp.shape(data_u)[0])
        self.noise_level = noise_level
        self.__dict__.pop('_prob', None)

    # ---- device state --------------------------------------------------------------------------------------------------
    def _drop_device_state(self):
        for name in ('_dTheta_u', '_dTheta_a', '_dL_u', '_dL_a'):
            a = self.__dict__.pop(name, None)
            if a is not None:
                a.free()
        p = self.__dict__.pop('_prob', None)
        if p is not None:
            p.free()                                           # (workspace, inverted blocks, prepared operators: everything but the factors)
        for name in ('_Theta_u_host', '_Theta_a_host', '_L_u_host', '_L_a_host'):
            self.__dict__.pop(name, None)

    def Gram_matrix(self, kernel='Gaussian', kernel_parameter=0.2, nugget=1e-10, nugget_type='adaptive'):
        if nugget_type not in ('adaptive', 'identity', 'none'):
            raise AttributeError(f"nugget_type {nugget_type!r}: the reference leaves Theta_u/Theta_a unset here")
        ctx = get_context()
        self._drop_device_state()
        self.nugget_type = nugget_type
        self.nugget = nugget
        self.kernel = kernel
        self.kernel_parameter = kernel_parameter
        self._dTheta_u, _ = ctx.assemble('Darcy_u', kernel, kernel_parameter, self.X_domain, self.X_boundary, nugget, nugget_type)
        self._dTheta_a, _ = ctx.assemble('Darcy_a', kernel, kernel_parameter, self.X_domain, self.X_boundary, nugget, nugget_type)

    def Gram_Cholesky(self):
        ctx = get_context()
        self._dL_u = self._dTheta_u.clone()
        self._dL_a = self._dTheta_a.clone()
        self.chol_info = (ctx.potrf(self._dL_u), ctx.potrf(self._dL_a))    # the reference has no guard here at all

    def _host(self, cache, dev, tril=False):
        if cache not in self.__dict__:
            d = self.__dict__.get(dev)
            if d is None:
                raise AttributeError(f'{cache[1:-5]}: not computed yet')
            a = d.download()
            self.__dict__[cache] = onp.tril(a) if tril else a
        return self.__dict__[cache]

    Theta_u = property(lambda self: self._host('_Theta_u_host', '_dTheta_u'))
    Theta_a = property(lambda self: self._host('_Theta_a_host', '_dTheta_a'))
    L_u = property(lambda self: self._host('_L_u_host', '_dL_u', True))
    L_a = property(lambda self: self._host('_L_a_host', '_dL_a', True))

    def _problem(self):
        if getattr(self, '_prob', None) is None:
            if getattr(self, '_dL_u', None) is None:
                raise RuntimeError('call Gram_matrix() and Gram_Cholesky() first')
            self._prob = gpk.GNProblem(get_context(), 'Darcy_flow2d', self.N_domain, self.N_boundary, self.rhs_f, self.bdy_g,
                                       self._dL_u, p0=float(self.noise_level), data_u=self.data_u, L2=self._dL_a,
                                       structured=int(os.environ.get('GPK_STRUCTURED', '0') or 0))   # (opt-in, round 6: gpk_gn_structured_prepare / 2: + gpk_gn_gram_prepare)
        return self._prob

    # loss / grad_loss inherited (device); GN_loss restated on the host for API parity (reference :126-147)
    def GN_loss(self, z, z_old):
        ctx = get_context()
        z = onp.asarray(z, float); zo = onp.asarray(z_old, float)
        Nd = self.N_domain
        w0o, w1o, w2o, v1o, v2o = zo[:Nd], zo[Nd:2 * Nd], zo[2 * Nd:3 * Nd], zo[4 * Nd:5 * Nd], zo[5 * Nd:6 * Nd]
        w0, w1, w2, v0, v1, v2 = (z[k * Nd:(k + 1) * Nd] for k in range(6))
        v3 = (-self.rhs_f) * (-onp.exp(-w0o)) * w0 + (-v1o) * w1 + (-v2o) * w2 + (-w1o) * v1 + (-w2o) * v2
        tot = 0.0
        for Ld, vec in ((self._dL_a, onp.concatenate((w1, w2, w0))), (self._dL_u, onp.concatenate((v1, v2, v3, v0, self.bdy_g)))):
            d = ctx.array(vec)
            ctx.trsm(Ld, d, trans=False, nrhs=1)
            t = d.download()
            tot += float(t @ t)
        return tot + (1 / self.noise_level ** 2) * float(onp.sum((v0[:self.N_data] - self.data_u) ** 2))

    def Hessian_GN(self, z, z_old):
        return self._hessian(z_old)

    def GN_method(self, max_iter=3, step_size=1, initial_sol='rdm', print_hist=True):
        sol = self._initial(initial_sol, 6 * self.N_domain)
        self.init_sol = sol
        sol = self._gn_iterate(self._problem(), sol, max_iter, step_size, print_hist, check_nan=False)
        Nd = self.N_domain
        w0, w1, w2, v0, v1, v2 = (sol[k * Nd:(k + 1) * Nd] for k in range(6))
        self.sol_vec_a = onp.concatenate((w1, w2, w0))
        v3 = -v1 * w1 - v2 * w2 + (-self.rhs_f) * onp.exp(-w0)
        self.sol_vec_u = onp.concatenate((v1, v2, v3, v0, self.bdy_g), axis=0)

    def extend_sol(self, X_test):
        ctx = get_context()
        X_test = onp.asarray(X_test, dtype=onp.float64)
        self.X_test = X_test
        self.N_test = X_test.shape[0]
        out = {}
        for tag, layout, Ld, vec in (('a', 'Darcy_a', self._dL_a, self.sol_vec_a), ('u', 'Darcy_u', self._dL_u, self.sol_vec_u)):
            coeff = ctx.array(vec)
            ctx.potrs(Ld, coeff, nrhs=1)
            out[tag] = ctx.extend(layout, self.kernel, self.kernel_parameter, X_test, self.X_domain, self.X_boundary, coeff).download()
        self.extended_sol_a = out['a']
        self.extended_sol_u = out['u']

"""END-TO-END parity at the sizes the metric is quoted on, with the ORACLE'S OWN factor (round 5, verdict item 2).

tests/test_gpu_fullsize_oracle.py hands the DEVICE's factor to the oracle and follows the device's trajectory step by step; that
checks every map of the path separately.  The north star states the chain: "results match the reference path on identical collocation
points (L2 solution error within 1e-6 relative, identical GN iterate count)".  Here the two chains never meet before the end:

    product:  the driver's own flow (main_*.py: parse -> seed -> solver_GP -> auto_sample -> solve -> test) on the device
    oracle:   O.gram_matrix_assembly + O.add_nugget -> numpy.linalg.cholesky (LAPACK dpotrf) -> O.gn_method (ALL steps of the
              reference configuration, from the same seeded start, differences compounding) -> O.construct_theta_test + O.extend
              (reference: src/solver.py:139-160,180-184; src/PDEs.py:56-135,203-208,250-350; src/InverseProblems.py:66-196)

asserted: `sol_sampled_pts` (the Gauss-Newton iterate) and `extended_sol` on the driver's test grid within 1e-6 RELATIVE (2-norm), the same
number of loss-history entries; printed: the deviations and both sides' L2 errors against the truth.  BASELINE configs 2, 3, 4 (config 5:
tests/test_gpu_targets.py; config 1: tests/test_gpu_parity.py against the reference-run fixture).  Burgers additionally against the
numbers stored in the reference's own notebook (produced by real JAX; unseeded there, hence statistical)."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')
if PKG not in sys.path:
    sys.path.insert(0, PKG)

TOL = 1e-6                                                        # north star: within 1e-6 relative of the reference path


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def _rms(a):
    return float(np.sqrt(np.mean(np.asarray(a) ** 2)))


def _oracle_chain(name, eqn, kernel, kp, Xd, Xb, nugget, make_system, z0, steps, Xt):
    """oracle Theta(s) -> LAPACK factor(s) -> all Gauss-Newton steps -> extension(s) on Xt; prints its own timing"""
    t0 = time.perf_counter()
    Nd, Nb = Xd.shape[0], Xb.shape[0]
    if eqn == 'Darcy_flow2d':
        Tu, Ta = O.gram_matrix_assembly(Xd, Xb, eqn, kernel, kp)
        Tu, _ = O.add_nugget(Tu, 'Darcy_u', Nd, Nb, nugget)
        Ta, _ = O.add_nugget(Ta, 'Darcy_a', Nd, Nb, nugget)
        Ls = [np.linalg.cholesky(Ta), np.linalg.cholesky(Tu)]     # oracle order [L_a, L_u]
        del Tu, Ta
    else:
        T, _ = O.add_nugget(O.gram_matrix_assembly(Xd, Xb, eqn, kernel, kp), eqn, Nd, Nb, nugget)
        Ls = [np.linalg.cholesky(T)]                              # raises LinAlgError if LAPACK finds the matrix indefinite
        del T
    t1 = time.perf_counter()
    sysm = make_system()
    sol, hist = O.gn_method(sysm, Ls, z0, steps, 1)
    t2 = time.perf_counter()
    sv = sysm.sol_vec(sol)
    if eqn == 'Darcy_flow2d':
        Ttu, Tta = O.construct_theta_test(Xt, Xd, Xb, eqn, kernel, kp)
        ext = (O.extend(Ls[0], Tta, sv[0]), O.extend(Ls[1], Ttu, sv[1]))          # (a, u)
    else:
        ext = O.extend(Ls[0], O.construct_theta_test(Xt, Xd, Xb, eqn, kernel, kp), sv[0])
    print(f'\n[{name}] oracle chain: Theta + LAPACK factor {t1 - t0:.1f} s, {steps} Gauss-Newton steps {t2 - t1:.1f} s, extension {time.perf_counter() - t2:.1f} s')
    return sol, hist, ext


def test_config2_end_to_end_elliptic():
    import main_NonLinElliptic2d as drv
    from _driver_common import solve_forward, tensor_grid
    cfg = drv.parse(['--N_domain', '4000', '--N_boundary', '400', '--print_hist', '', '--show_figure', ''])   # kernel, sigma 0.2, nugget 1e-13, 4 steps: defaults
    assert (cfg.kernel, cfg.kernel_parameter, cfg.nugget, cfg.GNsteps, cfg.initial_sol) == ('Gaussian', 0.2, 1e-13, 4, 'rdm')
    u, f = drv.manufactured(cfg.alpha, cfg.m)
    np.random.seed(0)
    s, _ = solve_forward(cfg, 'Nonlinear_elliptic', u, f, drv.UNIT_SQUARE, solve_kwargs={'method': 'elimination'}, verbose=False)
    e = s.eqn
    assert e.chol_info == 0 and e.X_domain.shape == (4000, 2) and e.X_boundary.shape == (400, 2)
    _, _, Xt = tensor_grid(60, *drv.UNIT_SQUARE)
    s.test(Xt, print_option=False)
    Xd, Xb = e.X_domain, e.X_boundary
    sol_o, hist_o, ext_o = _oracle_chain('C2', 'Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, cfg.nugget,
                                         lambda: O.EllipticSystem(cfg.alpha, cfg.m, f(Xd[:, 0], Xd[:, 1]), u(Xb[:, 0], Xb[:, 1])),
                                         e.init_sol, cfg.GNsteps, Xt)
    r_sol, r_ext = _rel(e.sol_sampled_pts, sol_o), _rel(e.extended_sol, ext_o)
    tru_p, tru_t = u(Xd[:, 0], Xd[:, 1]), u(Xt[:, 0], Xt[:, 1])
    print(f'[C2] end to end, device vs oracle (own LAPACK factor, {cfg.GNsteps} compounded steps): sol_sampled_pts rel. dev {r_sol:.2e}, '
          f'extended_sol rel. dev {r_ext:.2e}; pts_L2_err device {_rms(e.sol_sampled_pts - tru_p):.3e} oracle {_rms(sol_o - tru_p):.3e}; '
          f'test_L2_err device {_rms(e.extended_sol - tru_t):.3e} oracle {_rms(ext_o - tru_t):.3e}; '
          f'loss history device {["%.6e" % v for v in e.loss_hist]} oracle {["%.6e" % v for v in hist_o]}')
    assert len(e.loss_hist) == len(hist_o) == cfg.GNsteps + 1     # identical iterate count
    assert r_sol <= TOL and r_ext <= TOL
    # the loss at nugget 1e-13 is ||L^{-1}F||^2 with a marginally definite Theta (smallest pivots 3e-5 against entries 5e3): two equally
    # valid fp64 factorisations of matrices that agree to 4e-16 give values that differ in the second to third digit (SURVEY section 0;
    # first GPU run: J(z_0) 7.227e16 against 7.150e16 with LAPACK's factor) -- NOT a parity quantity at this nugget, bounded loosely; the
    # iterates above are (and the loss history IS asserted to 1e-6 where the nugget makes it meaningful: configs 3 and 4 below).  Measured:
    # history 7.227e16, 4.337e13, 4.965e8, 8559.7, 8558.6 against 7.150e16, 4.186e13, 5.267e8, 8563.7, 8562.0 with LAPACK's factor -- up to 6 %
    # apart on the way, 4e-4 at convergence -- while the iterates agree to 6.9e-9 and the extension to 8.7e-9
    np.testing.assert_allclose(e.loss_hist, hist_o, rtol=0.25)
    assert e.loss_hist[-1] == pytest.approx(hist_o[-1], rel=5e-3)


def test_config3_end_to_end_burgers():
    import main_Burgers1d as drv
    from _driver_common import seed_from, solve_forward, tensor_grid
    cfg = drv.parse(['--N_domain', '2000', '--N_boundary', '400', '--print_hist', '', '--show_figure', ''])   # sigma [0.3, 0.05], nugget 1e-5, 8 steps, seed 0: defaults
    assert (cfg.kernel, cfg.kernel_parameter, cfg.nugget, cfg.GNsteps, cfg.randomseed) == ('anisotropic_Gaussian', [0.3, 0.05], 1e-5, 8, 0)
    seed_from(cfg)
    s, _ = solve_forward(cfg, 'Burgers', drv.initial_and_lateral, lambda x1, x2: 0, drv.SPACE_TIME, verbose=False)
    e = s.eqn
    Xd, Xb = e.X_domain, e.X_boundary
    assert (Xd.shape[0], Xb.shape[0]) == (2000, 399) and e.chol_info == 0
    _, _, Xt = tensor_grid(60, *drv.SPACE_TIME)
    s.test(Xt, print_option=False)
    sol_o, hist_o, ext_o = _oracle_chain('C3', 'Burgers', 'anisotropic_Gaussian', cfg.kernel_parameter, Xd, Xb, cfg.nugget,
                                         lambda: O.BurgersSystem(cfg.alpha, cfg.nu, np.zeros(Xd.shape[0]), drv.initial_and_lateral(Xb[:, 0], Xb[:, 1])),
                                         e.init_sol, cfg.GNsteps, Xt)
    Nd = Xd.shape[0]
    r_sol, r_ext = _rel(e.sol_sampled_pts, sol_o[:Nd]), _rel(e.extended_sol, ext_o)
    truth = drv.cole_hopf_truth(cfg.nu)
    tru_p, tru_t = truth(Xd[:, 0], Xd[:, 1]), truth(Xt[:, 0], Xt[:, 1])
    print(f'[C3] end to end, device vs oracle (own LAPACK factor, {cfg.GNsteps} compounded steps): sol_sampled_pts rel. dev {r_sol:.2e}, '
          f'extended_sol rel. dev {r_ext:.2e}; pts_L2_err device {_rms(e.sol_sampled_pts - tru_p):.3e} oracle {_rms(sol_o[:Nd] - tru_p):.3e}; '
          f'test_L2_err device {_rms(e.extended_sol - tru_t):.3e} oracle {_rms(ext_o - tru_t):.3e}')
    assert len(e.loss_hist) == len(hist_o) == cfg.GNsteps + 1
    assert r_sol <= TOL and r_ext <= TOL
    np.testing.assert_allclose(e.loss_hist, hist_o, rtol=1e-6)    # nugget 1e-5: the loss history is a parity quantity here


def test_config4_end_to_end_darcy():
    import main_DarcyFlow2d as drv
    from _driver_common import seed_from, tensor_grid
    from scipy.interpolate import griddata
    from reference_solver.FD_for_Darcy_flow import FD_Darcy_flow_2d
    from src.solver import solver_GP
    cfg = drv.parse(['--N_domain', '1600', '--N_boundary', '200', '--N_data', '60', '--noise_level', '1e-3', '--print_hist', '', '--show_figure', ''])
    assert (cfg.kernel, cfg.kernel_parameter, cfg.nugget, cfg.GNsteps, cfg.randomseed) == ('Gaussian', 0.2, 1e-8, 8, 9999)
    seed_from(cfg)
    s = solver_GP(cfg, PDE_type='Darcy_flow2d')                   # main_DarcyFlow2d.main, step by step (it keeps no handle on the solver)
    s.set_equation(bdy=lambda x1, x2: 0, rhs=drv.source, domain=np.array(drv.UNIT_SQUARE), print_option=False)
    s.auto_sample_IP(cfg.N_domain, cfg.N_boundary, cfg.N_data, sampled_type=cfg.sampled_type, print_option=False)
    XX, YY, Xg = tensor_grid(drv.GRID, *drv.UNIT_SQUARE)
    u_grid = FD_Darcy_flow_2d(drv.GRID - 2, drv.permeability, drv.source)
    Xo = s.eqn.X_data
    observed = griddata((XX.flatten(), YY.flatten()), u_grid.reshape(-1, 1), (Xo[:, 0], Xo[:, 1]), method='linear')[:, 0]
    s.get_observed_data(observed, cfg.noise_level, print_option=False)
    s.solve(print_option=False)
    s.test(Xg, print_option=False)
    e = s.eqn
    Xd, Xb = e.X_domain, e.X_boundary
    Nd, Nb = Xd.shape[0], Xb.shape[0]
    sol_o, hist_o, (ext_a_o, ext_u_o) = _oracle_chain('C4', 'Darcy_flow2d', 'Gaussian', 0.2, Xd, Xb, cfg.nugget,
                                                      lambda: O.DarcySystem(np.ones(Nd), np.zeros(Nb), e.data_u, cfg.noise_level),
                                                      e.init_sol, cfg.GNsteps, Xg)
    sv_a_o, sv_u_o = O.DarcySystem(np.ones(Nd), np.zeros(Nb), e.data_u, cfg.noise_level).sol_vec(sol_o)
    r = {'sol_vec_u': _rel(e.sol_vec_u, sv_u_o), 'sol_vec_a': _rel(e.sol_vec_a, sv_a_o),
         'extended_sol_u': _rel(e.extended_sol_u, ext_u_o), 'extended_sol_a': _rel(e.extended_sol_a, ext_a_o)}
    a_true = drv.permeability(Xg[:, 0], Xg[:, 1])
    print(f'[C4] end to end, device vs oracle (own LAPACK factors, {cfg.GNsteps} compounded steps): ' + ', '.join(f'{k} rel. dev {v:.2e}' for k, v in r.items())
          + f'; u test_L2_err device {_rms(e.extended_sol_u - u_grid.reshape(-1)):.3e} oracle {_rms(ext_u_o - u_grid.reshape(-1)):.3e}; '
          f'a test_L2_err device {_rms(np.exp(e.extended_sol_a) - a_true):.3e} oracle {_rms(np.exp(ext_a_o) - a_true):.3e}')
    assert len(e.loss_hist) == len(hist_o) == cfg.GNsteps + 1
    assert all(v <= TOL for v in r.values()), r
    # loss history: nugget 1e-8 on two factors of order 6600 / 4800 and 8 compounded steps of an inverse problem (cond(H) ~ 1e11): the printed
    # deviation is what two fp64 chains give (iterates above: <= 7e-8; measured 2.7e-6 on the loss); bound 2e-5, the start value (no iteration involved) 1e-6
    dev = np.max(np.abs(np.asarray(e.loss_hist) / np.asarray(hist_o) - 1.0))
    print(f'[C4] loss history max rel. dev {dev:.2e} (start value {abs(e.loss_hist[0] / hist_o[0] - 1):.1e}); device {["%.8e" % v for v in e.loss_hist]}')
    assert e.loss_hist[0] == pytest.approx(hist_o[0], rel=1e-6)
    np.testing.assert_allclose(e.loss_hist, hist_o, rtol=2e-5)


def test_burgers_notebook_statistical_kat():
    """The reference's Burgers notebook (notebooks/Burgers_anisotropic_kernel.ipynb, cells 13 and 16: N_domain 1000, N_boundary 201 -> 3 x 67
    samples, anisotropic Gaussian sigma [3, 20] -- the notebook's kernel takes PRECISIONS' inverse scales, i.e. sigma = [1/3, 1/20] in the
    src/ convention -- nugget 1e-5, 12 Gauss-Newton steps from a N(0,1) start) is UNSEEDED, so its stored JAX output is a sample: final loss
    ~24.8, L2 error ~4e-3 (stored digits: tests/test_oracle_kat.py holds the exact strings).  The device run from seed 0 must land in the same
    regime (the CPU twin tests/test_oracle_kat.py::test_burgers_notebook_statistical calibrated seeds 0-2: final J 25.14-25.27, L2 error
    3.6e-3-6.2e-3): final loss within 3 % of the stored one, L2 error against the Cole-Hopf truth in [2e-3, 1e-2] -- and agree with the
    oracle chain from the same seed to 1e-6 (which pins the Burgers block placement against the oracle on a THIRD problem size)."""
    import gpk                                                     # noqa: F401
    import main_Burgers1d as drv
    from _driver_common import solve_forward, tensor_grid
    cfg = drv.parse(['--N_domain', '1000', '--N_boundary', '201', '--GNsteps', '12', '--kernel_parameter', str(1 / 3), str(1 / 20),
                     '--print_hist', '', '--show_figure', ''])
    np.random.seed(0)
    s, _ = solve_forward(cfg, 'Burgers', drv.initial_and_lateral, lambda x1, x2: 0, drv.SPACE_TIME, verbose=False)
    e = s.eqn
    Xd, Xb = e.X_domain, e.X_boundary
    assert (Xd.shape[0], Xb.shape[0]) == (1000, 201)
    _, _, Xt = tensor_grid(60, *drv.SPACE_TIME)
    s.test(Xt, print_option=False)
    truth = drv.cole_hopf_truth(cfg.nu)
    l2 = _rms(e.extended_sol - truth(Xt[:, 0], Xt[:, 1]))
    print(f'\n[Burgers notebook KAT] device from seed 0: final loss {e.loss_hist[-1]:.4f} (notebook sample: ~24.8), test L2 error {l2:.3e} (notebook sample: ~4e-3)')
    assert all(np.isfinite(e.loss_hist)) and e.loss_hist[-1] < e.loss_hist[0]
    assert e.loss_hist[-1] == pytest.approx(24.822667800854497, rel=0.03)      # stored final J of the notebook's sample
    assert 2e-3 < l2 < 1e-2                                                    # stored space-time L2 error: 0.0040087824448355285
    sol_o, hist_o, ext_o = _oracle_chain('Burgers notebook', 'Burgers', 'anisotropic_Gaussian', cfg.kernel_parameter, Xd, Xb, cfg.nugget,
                                         lambda: O.BurgersSystem(cfg.alpha, cfg.nu, np.zeros(1000), drv.initial_and_lateral(Xb[:, 0], Xb[:, 1])),
                                         e.init_sol, cfg.GNsteps, Xt)
    assert _rel(e.sol_sampled_pts, sol_o[:1000]) <= TOL and _rel(e.extended_sol, ext_o) <= TOL
    np.testing.assert_allclose(e.loss_hist, hist_o, rtol=1e-6)

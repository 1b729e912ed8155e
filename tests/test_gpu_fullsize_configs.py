"""BASELINE configs 3 and 4 at their FULL sizes on the GPU, through the reference's class API (solver_GP):
    C3  Burgers, anisotropic_Gaussian sigma = [0.3, 0.05], N_domain = 2000, N_boundary = 400 (-> 399), 8 GN steps, seed 0
        (Theta of order 8399, 6000 unknowns)
    C4  Darcy flow inverse problem, Gaussian sigma = 0.2, N_domain = 1600, N_boundary = 200, N_data = 60, noise 1e-3,
        8 GN steps, seed 9999 (Theta_u 6600, Theta_a 4800, 9600 unknowns)
The oracle needs minutes per Gauss-Newton step at these sizes, so the checks are size-independent properties
(SURVEY 8c): the loss history decreases, the solution reaches the independent truth (Cole-Hopf quadrature / finite
differences) to the accuracy the method has at this resolution, and the two solve schedules of the library -- GEMM-only
solve through the inverted diagonal blocks of the factor (default) and true substitution -- give the same iterate."""
import argparse
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def _set(lib, **kv):
    for k, v in kv.items():
        lib.gpk_debug_set(int(k[1:]), int(v))


def _burgers(steps=8):
    from src.solver import solver_GP
    cfg = argparse.Namespace(alpha=1.0, nu=0.02, kernel='anisotropic_Gaussian', kernel_parameter=[0.3, 0.05], nugget=1e-5,
                             nugget_type='adaptive', GNsteps=steps, step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(0)
    s = solver_GP(cfg, PDE_type='Burgers')
    s.set_equation(bdy=lambda x1, x2: -np.sin(np.pi * x2) * (x1 == 0) + 0 * (x2 == 0), rhs=lambda x1, x2: 0,
                   domain=np.array([[0, 1], [-1, 1]]), print_option=False)
    s.auto_sample(2000, 400, print_option=False)
    t0 = time.perf_counter()
    s.solve(print_option=False)
    return s, time.perf_counter() - t0


def test_burgers_config3_full_size():
    import gpk
    from main_Burgers1d import cole_hopf_truth
    lib = gpk.load_library()
    s, secs = _burgers()
    e = s.eqn
    assert e.N_domain == 2000 and e.N_boundary == 399
    hist = np.asarray(e.loss_hist)
    assert hist.shape == (9,) and np.all(np.isfinite(hist))
    assert np.all(hist[1:] <= hist[:-1] * (1 + 1e-9))            # monotone decrease
    assert abs(hist[-1] - hist[-2]) <= 1e-8 * hist[-1]           # converged in 8 steps
    assert 20.0 < hist[-1] < 40.0                                # the reference's notebook run (unseeded) ends near 25
    # accuracy against the Cole-Hopf truth on the drivers' 60 x 60 test grid
    a = np.linspace(0, 1, 60); b = np.linspace(-1, 1, 60)
    XX, YY = np.meshgrid(a, b)
    Xt = np.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
    s.test(Xt, print_option=False)
    truth = cole_hopf_truth(0.02)(Xt[:, 0], Xt[:, 1])
    l2 = np.sqrt(np.mean((e.extended_sol - truth) ** 2))
    assert l2 < 2e-3, l2
    truth_pts = cole_hopf_truth(0.02)(e.X_domain[:, 0], e.X_domain[:, 1])
    assert np.sqrt(np.mean((e.sol_sampled_pts - truth_pts) ** 2)) < 2e-3
    print(f'\n[C3] 8 Gauss-Newton steps + assembly + Cholesky: {secs * 1e3:.0f} ms, final loss {hist[-1]:.6f}, test L2 {l2:.3e}')
    # same problem with true substitution in the solve S = L^{-1}[A | F]: same iterate
    sol = np.array(e.sol_sampled_pts)
    try:
        _set(lib, k10=0)
        s2, _ = _burgers()
    finally:
        _set(lib, k10=1)
    assert _rel(s2.eqn.sol_sampled_pts, sol) < 1e-8
    np.testing.assert_allclose(s2.eqn.loss_hist, hist, rtol=1e-6)


def test_darcy_config4_full_size():
    import gpk
    from scipy.interpolate import griddata
    from main_DarcyFlow2d import permeability, source
    from reference_solver.FD_for_Darcy_flow import FD_Darcy_flow_2d
    from src.solver import solver_GP
    lib = gpk.load_library()
    GRID = 80
    u_grid = FD_Darcy_flow_2d(GRID - 2, permeability, source)
    xx = np.linspace(0, 1, GRID)
    XX, YY = np.meshgrid(xx, xx)
    Xg = np.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)

    def run():
        cfg = argparse.Namespace(kernel='Gaussian', kernel_parameter=0.2, nugget=1e-8, nugget_type='adaptive', GNsteps=8,
                                 step_size=1, initial_sol='rdm', print_hist=False)
        np.random.seed(9999)
        s = solver_GP(cfg, PDE_type='Darcy_flow2d')
        s.set_equation(bdy=lambda x1, x2: 0, rhs=source, domain=np.array([[0, 1], [0, 1]]), print_option=False)
        s.auto_sample_IP(1600, 200, 60, print_option=False)
        Xo = s.eqn.X_data
        obs = griddata((XX.flatten(), YY.flatten()), u_grid.reshape(-1, 1), (Xo[:, 0], Xo[:, 1]), method='linear')[:, 0]
        s.get_observed_data(obs, 1e-3, print_option=False)
        t0 = time.perf_counter()
        s.solve(print_option=False)
        return s, time.perf_counter() - t0

    s, secs = run()
    e = s.eqn
    assert e.N_domain == 1600 and e.N_boundary == 200 and e.N_data == 60
    hist = np.asarray(e.loss_hist)
    assert hist.shape == (9,) and np.all(np.isfinite(hist))
    assert np.all(hist[1:] < hist[:-1])                          # still descending after 8 steps (as in the reference's runs)
    assert hist[-1] < 1e-6 * hist[0]
    s.test(Xg, print_option=False)
    u_gp = np.reshape(e.extended_sol_u, (GRID, GRID))
    l2u = np.sqrt(np.mean((u_gp - u_grid) ** 2))
    assert l2u < 2e-3, l2u                                       # the recovered state matches the finite-difference truth
    # the observations are reproduced to the noise level
    assert np.sqrt(np.mean((e.sol_vec_u[3 * 1600:3 * 1600 + 60] - e.data_u) ** 2)) < 5e-3
    print(f'\n[C4] 8 Gauss-Newton steps + 2 assemblies + 2 Cholesky: {secs * 1e3:.0f} ms, final loss {hist[-1]:.4f}, u test L2 {l2u:.3e}')
    sol_a, sol_u = np.array(e.sol_vec_a), np.array(e.sol_vec_u)
    try:
        _set(lib, k10=0)
        s2, _ = run()
    finally:
        _set(lib, k10=1)
    # (inverse problem, 8 steps, not converged: differences between the schedules are amplified along the way)
    assert _rel(s2.eqn.sol_vec_u, sol_u) < 1e-6 and _rel(s2.eqn.sol_vec_a, sol_a) < 1e-6
    np.testing.assert_allclose(s2.eqn.loss_hist, hist, rtol=1e-5)

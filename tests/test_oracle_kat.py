"""Known-answer tests: the oracle against numbers COMMITTED IN THE REFERENCE (stored notebook outputs).

These pin the CPU oracle (oracle/gp_oracle.py) without JAX: each test replays a notebook's RNG recipe on
numpy's legacy global generator and compares with the notebook's stored stdout (tests/nb_recipes.py).
Tolerances: loss histories rtol 1e-8 (Elliptic, Darcy; observed ~3e-11) and 1e-6 (Eikonal; observed ~3e-10,
worse conditioning), trace ratios rtol 1e-14, errors rtol 1e-7.
"""
import numpy as np
import pytest

from oracle import gp_oracle as O
from oracle import truth_solvers as TS
import nb_recipes as R


@pytest.fixture(scope='module')
def elliptic_run():
    np.random.seed(10)                                            # Nonlinear_Elliptic_Equation.ipynb c2:9
    Xd, Xb = R.notebook_sample_points(900, 124)                   # c4:26-29 (N_pts=30)
    T = O.gram_matrix_assembly(Xd, Xb, 'Nonlinear_elliptic', 'Gaussian', 0.2)
    Tl, ratio = O.add_nugget(T, 'Nonlinear_elliptic', 900, 124, 1e-4)     # c9:44 set_nugget = 1e-4
    L = O.cholesky(Tl)
    init = np.random.normal(0.0, 1.0, 900)                        # c9:40
    sysm = O.EllipticSystem(1, 3, O.elliptic_rhs(Xd[:, 0], Xd[:, 1], 1, 3), O.elliptic_truth(Xb[:, 0], Xb[:, 1]))
    sol, hist = O.gn_method(sysm, [L], init, 5, 1)
    return dict(Xd=Xd, Xb=Xb, L=L, ratio=ratio, sol=sol, hist=hist, sysm=sysm)


def test_elliptic_trace_ratio(elliptic_run):
    assert elliptic_run['ratio'][0] == pytest.approx(R.ELLIPTIC_RATIO, rel=1e-14)


def test_elliptic_loss_history(elliptic_run):
    np.testing.assert_allclose(elliptic_run['hist'], R.ELLIPTIC_J, rtol=1e-8)


def test_elliptic_collocation_errors(elliptic_run):
    Xd, sol = elliptic_run['Xd'], elliptic_run['sol']
    err = np.abs(O.elliptic_truth(Xd[:, 0], Xd[:, 1]) - sol)
    assert np.sqrt(np.sum(err ** 2) / 900) == pytest.approx(R.ELLIPTIC_PTS_L2, rel=1e-7)
    assert err.max() == pytest.approx(R.ELLIPTIC_PTS_MAX, rel=1e-7)


def test_elliptic_test_grid_errors(elliptic_run):
    """100x100 test grid of c12: L2 = ||.||_F / num_pts, max."""
    n = 100
    xx = np.linspace(0, 1, n)
    XX, YY = np.meshgrid(xx, xx)
    Xt = np.stack([XX.ravel(), YY.ravel()], axis=1)
    r = elliptic_run
    Tt = O.construct_theta_test(Xt, r['Xd'], r['Xb'], 'Nonlinear_elliptic', 'Gaussian', 0.2)
    ext = O.extend(r['L'], Tt, r['sysm'].sol_vec(r['sol'])[0]).reshape(n, n)
    truth = O.elliptic_truth(XX, YY)
    assert np.linalg.norm(ext - truth, 'fro') / n == pytest.approx(R.ELLIPTIC_TEST_L2, rel=1e-7)
    assert np.abs(ext - truth).max() == pytest.approx(R.ELLIPTIC_TEST_MAX, rel=1e-6)


def test_elliptic_reference_op_sequence_agrees(elliptic_run):
    """The reference's general-LU operation sequence (B1) and the triangular formulation (B2) give the same
    iterates at this nugget (1e-4): loss history to 1e-9."""
    r = elliptic_run
    np.random.seed(10)
    R.notebook_sample_points(900, 124)
    init = np.random.normal(0.0, 1.0, 900)
    _, hist = O.gn_method(r['sysm'], [r['L']], init, 2, 1, faithful=True)
    np.testing.assert_allclose(hist, r['hist'][:3], rtol=1e-9)


def test_darcy_notebook():
    from scipy.interpolate import griddata
    np.random.seed(10)                                            # Darcy_flow_IP_noisy.ipynb c2:2
    u_true = TS.fd_darcy_flow_2d(100, R.darcy_a, lambda x, y: 1.0 + 0 * x)        # c4: plot_u(100), no RNG use
    xx = np.linspace(0, 1, 102)
    XX, YY = np.meshgrid(xx, xx)
    Xd, Xb = R.notebook_sample_points(400, 100)                   # c5
    Tu, Ta = O.gram_matrix_assembly(Xd, Xb, 'Darcy_flow2d', 'Gaussian', 0.2)
    Tul, ru = O.add_nugget(Tu, 'Darcy_u', 400, 100, 1e-5)
    Tal, ra = O.add_nugget(Ta, 'Darcy_a', 400, 100, 1e-5)
    np.testing.assert_allclose(ru, R.DARCY_RATIO_U, rtol=1e-14)
    np.testing.assert_allclose(ra, R.DARCY_RATIO_A, rtol=1e-14)
    Lu, La = O.cholesky(Tul), O.cholesky(Tal)
    init = np.random.normal(0, 1.0, 2400)                         # c10:84, drawn BEFORE the noise (c10:60)
    data = griddata((XX.ravel(), YY.ravel()), u_true.reshape(-1, 1), (Xd[:40, 0], Xd[:40, 1]), method='linear')[:, 0]
    data = data + 1e-3 * np.random.normal(0, 1.0, 40)
    sysm = O.DarcySystem(np.ones(400), np.zeros(100), data, 1e-3)
    _, hist = O.gn_method(sysm, [La, Lu], init, 8, 1)
    np.testing.assert_allclose(hist, R.DARCY_J, rtol=1e-8)


def test_eikonal_notebook():
    np.random.seed(20)                                            # Regularized_Eikonal...ipynb c2:2
    Xd, Xb = R.notebook_sample_points(400, 84)                    # c5 (N_pts=20)
    T = O.gram_matrix_assembly(Xd, Xb, 'Eikonal', 'Gaussian', 0.2)
    Tl, ratio = O.add_nugget(T, 'Eikonal', 400, 84, 1e-6)         # c9 set_nugget = 1e-6 (global read in c8 body)
    np.testing.assert_allclose(ratio, R.EIKONAL_RATIO, rtol=1e-14)
    L = O.cholesky(Tl)
    sysm = O.EikonalSystem(1e-2, np.ones(400), np.zeros(84))
    sol, hist = O.gn_method(sysm, [L], np.zeros(1200), 10, 1.0)
    np.testing.assert_allclose(hist, R.EIKONAL_J, rtol=1e-6)
    # test error on the 100x100 interior grid vs the Cole-Hopf FD truth (c4, c11-c12, c15)
    n = 100
    XX, YY, truth = TS.cole_hopf_eikonal(n, 1e-2)
    Xt = np.stack([XX.ravel(), YY.ravel()], axis=1)
    Tt = O.construct_theta_test(Xt, Xd, Xb, 'Eikonal', 'Gaussian', 0.2)
    ext = O.extend(L, Tt, sysm.sol_vec(sol)[0]).reshape(n, n)
    assert np.linalg.norm(ext - truth, 'fro') / n == pytest.approx(R.EIKONAL_TEST_L2, rel=1e-5)
    assert np.abs(ext - truth).max() == pytest.approx(R.EIKONAL_TEST_MAX, rel=1e-4)


def test_cole_hopf_known_value():
    """reference Cole_Hopf solve_Eikonal(58, 0.1) -> max u = 0.366745974372574."""
    _, _, u = TS.cole_hopf_eikonal(58, 0.1)
    assert u.shape == (58, 58)
    assert u.max() == pytest.approx(0.366745974372574, rel=1e-10)

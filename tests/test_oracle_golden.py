"""Oracle vs fixtures produced by executing the reference's own src/ (tests/golden/gen/make_fixtures.py).

theta_small.npz : Gram_matrix_assembly / construct_Theta_test outputs for the four layouts x both kernel classes
solves.npz      : complete solver_GP.solve() + test() runs (points, rhs, init, loss_hist, sol, extended_sol)
sampling.npz    : src/sample_points.py outputs (imported directly, numpy only)
Tolerances: Theta 2e-15 of the block scale; loss histories 1e-5 (nugget 1e-8) to 1e-7 (nugget >= 1e-6) relative (autodiff+LU reference
vs closed-form+triangular oracle differ by cond*eps); solution vectors 1e-6 (the north-star parity bound);
BASELINE config 1 (nugget 1e-13): solution 1e-6, loss history NOT compared.
"""
import os

import numpy as np
import pytest

from oracle import gp_oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')
EQN = {'elliptic': 'Nonlinear_elliptic', 'burgers': 'Burgers', 'eikonal': 'Eikonal', 'darcy': 'Darcy_flow2d'}


@pytest.fixture(scope='module')
def theta():
    return np.load(os.path.join(G, 'theta_small.npz'))


@pytest.fixture(scope='module')
def solves():
    return np.load(os.path.join(G, 'solves.npz'))


def _cases(d):
    return sorted({k.split('__')[0] for k in d.files})


def _kp(name, d):
    kp = d[name + '__kp']
    return ('Gaussian', float(kp[0])) if name.endswith('gauss') else ('anisotropic_Gaussian', [float(kp[0]), float(kp[1])])


def test_theta_blocks(theta):
    for name in _cases(theta):
        eqn = EQN[name.split('_')[0]]
        kernel, kp = _kp(name, theta)
        Xd, Xb, Xt = theta[name + '__Xd'], theta[name + '__Xb'], theta[name + '__Xt']
        T = O.gram_matrix_assembly(Xd, Xb, eqn, kernel, kp)
        Tt = O.construct_theta_test(Xt, Xd, Xb, eqn, kernel, kp)
        if eqn == 'Darcy_flow2d':
            pairs = [(T[0], 'Theta_u'), (T[1], 'Theta_a'), (Tt[0], 'Theta_u_test'), (Tt[1], 'Theta_a_test')]
        else:
            pairs = [(T, 'Theta'), (Tt, 'Theta_test')]
        for got, key in pairs:
            want = theta[f'{name}__{key}']
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= 2e-15 * np.max(np.abs(want)), (name, key)
            if key.startswith('Theta') and not key.endswith('test'):
                assert np.array_equal(want, want.T)       # reference output is exactly symmetric


def test_sampling_bit_exact():
    d = np.load(os.path.join(G, 'sampling.npz'))
    for name in _cases(d):
        nd, nb, a, b, c, e, td, seed = d[name + '__args']
        dom = np.array([[a, b], [c, e]])
        if name.startswith('grid'):
            Xd, Xb = O.sampled_pts_grid(int(nd), int(nb), dom, bool(td))
        else:
            np.random.seed(int(seed))
            Xd, Xb = O.sampled_pts_rdm(int(nd), int(nb), dom, bool(td))
            assert np.array_equal(np.random.uniform(0, 1, 3), d[name + '__tail'])
        assert np.array_equal(Xd, d[name + '__Xd']), name
        assert np.array_equal(Xb, d[name + '__Xb']), name


def test_sampling_kat_survey():
    """SURVEY §8a-S tiny KAT: seed(0), sampled_pts_rdm(5, 8, [[0,1],[0,1]])."""
    np.random.seed(0)
    Xd, Xb = O.sampled_pts_rdm(5, 8, np.array([[0, 1], [0, 1]]))
    np.testing.assert_allclose(Xd[:, 0], [0.5488135, 0.71518937, 0.60276338, 0.54488318, 0.4236548], atol=5e-9)
    np.testing.assert_allclose(Xb[:, 0], [0.79172504, 0.52889492, 1, 1, 0.07103606, 0.0871293, 0, 0], atol=5e-9)


def _run(sysm, Ls, init, steps, want_hist, rtol):
    sol, hist = O.gn_method(sysm, Ls, init, steps, 1)
    np.testing.assert_allclose(hist, want_hist, rtol=rtol)
    return sol


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _factor(Xd, Xb, eqn, layout, kernel, kp, nugget):
    T = O.gram_matrix_assembly(Xd, Xb, eqn, kernel, kp)
    Tl, ratio = O.add_nugget(T, layout, Xd.shape[0], Xb.shape[0], nugget)
    return O.cholesky(Tl), ratio


def test_elliptic_small(solves):
    d, p = solves, 'elliptic_small'
    alpha, m, sigma, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    np.testing.assert_allclose(d[p + '__rhs_f'], O.elliptic_rhs(Xd[:, 0], Xd[:, 1], alpha, m), rtol=1e-12, atol=1e-10)
    L, ratio = _factor(Xd, Xb, 'Nonlinear_elliptic', 'Nonlinear_elliptic', 'Gaussian', sigma, nug)
    np.testing.assert_allclose(ratio, d[p + '__ratio'], rtol=1e-14)
    sysm = O.EllipticSystem(alpha, m, d[p + '__rhs_f'], d[p + '__bdy_g'])
    sol = _run(sysm, [L], d[p + '__init_sol'], int(steps), d[p + '__loss_hist'], 1e-5)
    assert _rel(sol, d[p + '__sol']) < 1e-8
    Tt = O.construct_theta_test(d[p + '__X_test'], Xd, Xb, 'Nonlinear_elliptic', 'Gaussian', sigma)
    assert _rel(O.extend(L, Tt, sysm.sol_vec(sol)[0]), d[p + '__extended_sol']) < 1e-8


def test_elliptic_baseline_config1(solves):
    """BASELINE config 1 (900/124, sigma 0.2, nugget 1e-13, 4 steps, seed 0): solution-vector parity <= 1e-6."""
    d, p = solves, 'elliptic_c1'
    alpha, m, sigma, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor(Xd, Xb, 'Nonlinear_elliptic', 'Nonlinear_elliptic', 'Gaussian', sigma, nug)
    sysm = O.EllipticSystem(alpha, m, d[p + '__rhs_f'], d[p + '__bdy_g'])
    sol, hist = O.gn_method(sysm, [L], d[p + '__init_sol'], int(steps), 1)
    assert _rel(sol, d[p + '__sol']) < 1e-6
    Tt = O.construct_theta_test(d[p + '__X_test'], Xd, Xb, 'Nonlinear_elliptic', 'Gaussian', sigma)
    assert _rel(O.extend(L, Tt, sysm.sol_vec(sol)[0]), d[p + '__extended_sol']) < 1e-6
    truth = O.elliptic_truth(Xd[:, 0], Xd[:, 1])
    l2 = np.sqrt(np.sum((truth - sol) ** 2) / Xd.shape[0])
    l2_ref = np.sqrt(np.sum((truth - d[p + '__sol']) ** 2) / Xd.shape[0])
    assert l2 < 1e-6 and abs(l2 - l2_ref) < 2e-8


def test_elliptic_relaxed(solves):
    d, p = solves, 'elliptic_relaxed'
    alpha, m, sigma, nug, steps, _, lam = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor(Xd, Xb, 'Nonlinear_elliptic', 'Nonlinear_elliptic', 'Gaussian', sigma, nug)
    sysm = O.EllipticRelaxedSystem(alpha, m, d[p + '__rhs_f'], d[p + '__bdy_g'], lam)
    z = _run(sysm, [L, None], d[p + '__init_sol'], int(steps), d[p + '__loss_hist'], 1e-6)
    assert _rel(z[Xd.shape[0]:], d[p + '__sol']) < 1e-6
    assert _rel(sysm.sol_vec(z)[0], d[p + '__sol_vec']) < 1e-6


def test_burgers_small(solves):
    d, p = solves, 'burgers_small'
    alpha, nu, st, sx, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, ratio = _factor(Xd, Xb, 'Burgers', 'Burgers', 'anisotropic_Gaussian', [st, sx], nug)
    np.testing.assert_allclose(ratio, d[p + '__ratio'], rtol=1e-13)
    sysm = O.BurgersSystem(alpha, nu, d[p + '__rhs_f'], d[p + '__bdy_g'])
    z = _run(sysm, [L], d[p + '__init_sol'], int(steps), d[p + '__loss_hist'], 1e-7)
    assert _rel(z[:Xd.shape[0]], d[p + '__sol']) < 1e-7
    assert _rel(sysm.sol_vec(z)[0], d[p + '__sol_vec']) < 1e-7
    Tt = O.construct_theta_test(d[p + '__X_test'], Xd, Xb, 'Burgers', 'anisotropic_Gaussian', [st, sx])
    assert _rel(O.extend(L, Tt, sysm.sol_vec(z)[0]), d[p + '__extended_sol']) < 1e-7


def test_eikonal_small(solves):
    d, p = solves, 'eikonal_small'
    eps, sigma, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor(Xd, Xb, 'Eikonal', 'Eikonal', 'Gaussian', sigma, nug)
    sysm = O.EikonalSystem(eps, d[p + '__rhs_f'], d[p + '__bdy_g'])
    z = _run(sysm, [L], d[p + '__init_sol'], int(steps), d[p + '__loss_hist'], 1e-7)
    assert _rel(z[:Xd.shape[0]], d[p + '__sol']) < 1e-7
    Tt = O.construct_theta_test(d[p + '__X_test'], Xd, Xb, 'Eikonal', 'Gaussian', sigma)
    assert _rel(O.extend(L, Tt, sysm.sol_vec(z)[0]), d[p + '__extended_sol']) < 1e-7


def test_darcy_small(solves):
    d, p = solves, 'darcy_small'
    sigma, nug, steps, _, ndata, noise = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    Nd, Nb = Xd.shape[0], Xb.shape[0]
    Tu, Ta = O.gram_matrix_assembly(Xd, Xb, 'Darcy_flow2d', 'Gaussian', sigma)
    Lu = O.cholesky(O.add_nugget(Tu, 'Darcy_u', Nd, Nb, nug)[0])
    La = O.cholesky(O.add_nugget(Ta, 'Darcy_a', Nd, Nb, nug)[0])
    sysm = O.DarcySystem(d[p + '__rhs_f'], d[p + '__bdy_g'], d[p + '__data_u'], noise)
    z = _run(sysm, [La, Lu], d[p + '__init_sol'], int(steps), d[p + '__loss_hist'], 1e-7)
    sa, su = sysm.sol_vec(z)
    assert _rel(sa, d[p + '__sol_vec_a']) < 1e-7 and _rel(su, d[p + '__sol_vec_u']) < 1e-7
    Ttu, Tta = O.construct_theta_test(d[p + '__X_test'], Xd, Xb, 'Darcy_flow2d', 'Gaussian', sigma)
    assert _rel(O.extend(La, Tta, sa), d[p + '__extended_sol_a']) < 1e-7
    assert _rel(O.extend(Lu, Ttu, su), d[p + '__extended_sol_u']) < 1e-7

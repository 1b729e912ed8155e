"""End-to-end GPU tests THROUGH THE REFERENCE-SHAPED API (src.solver.solver_GP, src.PDEs, src.InverseProblems,
src.Gram_matrice): the same call sequence as the reference's main_*.py, compared with fixtures produced by running the
reference's own classes (tests/golden/solves.npz).  Bound: solution / extension vectors within 1e-6 relative
(north-star), loss histories within 1e-5 at nugget >= 1e-8."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.fixture(scope='module')
def solves():
    return np.load(os.path.join(G, 'solves.npz'))


def _u(x1, x2):
    return np.sin(np.pi * x1) * np.sin(np.pi * x2) + 2 * np.sin(4 * np.pi * x1) * np.sin(4 * np.pi * x2)


def _f(alpha, m):
    return lambda x1, x2: (2 * np.pi ** 2 * np.sin(np.pi * x1) * np.sin(np.pi * x2)
                           + 64 * np.pi ** 2 * np.sin(4 * np.pi * x1) * np.sin(4 * np.pi * x2) + alpha * _u(x1, x2) ** m)


def test_gram_matrix_assembly_function(solves):
    from src.Gram_matrice import Gram_matrix_assembly, construct_Theta_test
    d = np.load(os.path.join(G, 'theta_small.npz'))
    n = 'darcy_gauss'
    Tu, Ta = Gram_matrix_assembly(d[n + '__Xd'], d[n + '__Xb'], eqn='Darcy_flow2d', kernel='Gaussian', kernel_parameter=0.2)
    assert np.max(np.abs(Tu - d[n + '__Theta_u'])) <= 4e-15 * np.max(np.abs(Tu))
    assert np.max(np.abs(Ta - d[n + '__Theta_a'])) <= 4e-15 * np.max(np.abs(Ta))
    Ttu, Tta = construct_Theta_test(d[n + '__Xt'], d[n + '__Xd'], d[n + '__Xb'], eqn='Darcy_flow2d')
    assert np.max(np.abs(Ttu - d[n + '__Theta_u_test'])) <= 4e-15 * np.max(np.abs(Ttu))
    assert Gram_matrix_assembly(d[n + '__Xd'], d[n + '__Xb'], eqn='nonsense') is None       # reference falls through


def test_elliptic_through_solver_gp(solves, capsys):
    from src.solver import solver_GP
    d, p = solves, 'elliptic_small'
    alpha, m, sigma, nug, steps, seed = d[p + '__params']
    cfg = argparse.Namespace(alpha=alpha, m=m, kernel='Gaussian', kernel_parameter=sigma, nugget=nug, nugget_type='adaptive',
                             GNsteps=int(steps), step_size=1, initial_sol='rdm', print_hist=True)
    np.random.seed(int(seed))                                   # same seed as the reference run -> identical points
    s = solver_GP(cfg, PDE_type='Nonlinear_elliptic')
    s.set_equation(bdy=_u, rhs=_f(alpha, m), domain=np.array([[0, 1], [0, 1]]))
    s.auto_sample(300, 60, sampled_type='random')
    assert np.array_equal(s.eqn.X_domain, d[p + '__X_domain']) and np.array_equal(s.eqn.X_boundary, d[p + '__X_boundary'])
    s.solve(method='elimination')
    np.testing.assert_array_equal(s.eqn.init_sol, d[p + '__init_sol'])         # RNG consumed in the reference's order
    np.testing.assert_allclose(s.eqn.loss_hist, d[p + '__loss_hist'], rtol=1e-5)
    assert s.eqn.ratio == pytest.approx(float(d[p + '__ratio'][0]), rel=1e-13)
    assert _rel(s.eqn.sol_sampled_pts, d[p + '__sol']) < 1e-6
    assert _rel(s.eqn.sol_vec, d[p + '__sol_vec']) < 1e-6
    s.collocation_pts_err(_u(s.eqn.X_domain[:, 0], s.eqn.X_domain[:, 1]))
    s.test(d[p + '__X_test'])
    assert _rel(s.eqn.extended_sol, d[p + '__extended_sol']) < 1e-6
    s.get_test_error(_u(d[p + '__X_test'][:, 0], d[p + '__X_test'][:, 1]))
    out = capsys.readouterr().out
    for line in ('[Equation type] Nonlinear elliptic equation', '[Kernel] Gaussian', '[Gauss Newton] elimination approaches',
                 'iter = 0 Loss =', 'Gauss-Newton step size = 1', '[Collocation point error] L2 error', '[Test error] Max error'):
        assert line in out, line
    # attribute surface
    L = s.eqn.L
    assert L.shape == (660, 660) and np.allclose(np.triu(L, 1), 0)
    assert np.linalg.norm(L @ L.T - s.eqn.Theta) <= 1e-12 * np.linalg.norm(s.eqn.Theta)
    # loss / grad / Hessian API
    z = d[p + '__sol']
    assert s.eqn.loss(z) == pytest.approx(d[p + '__loss_hist'][-1], rel=1e-5)
    H = s.eqn.Hessian_GN(z, z); g = s.eqn.grad_loss(z)
    assert H.shape == (300, 300) and g.shape == (300,) and np.array_equal(H, H.T)
    assert s.eqn.GN_loss(z, z) > 0


def test_elliptic_relaxation_through_solver_gp(solves):
    from src.solver import solver_GP
    d, p = solves, 'elliptic_relaxed'
    alpha, m, sigma, nug, steps, seed, lam = d[p + '__params']
    cfg = argparse.Namespace(alpha=alpha, m=m, kernel='Gaussian', kernel_parameter=sigma, nugget=nug, nugget_type='adaptive',
                             GNsteps=int(steps), step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(int(seed))
    s = solver_GP(cfg, PDE_type='Nonlinear_elliptic')
    s.set_equation(bdy=_u, rhs=_f(alpha, m), domain=np.array([[0, 1], [0, 1]]), print_option=False)
    s.auto_sample(120, 40, print_option=False)
    s.solve(method='relaxation', pen_lambda=lam, print_option=False)
    np.testing.assert_allclose(s.eqn.loss_hist, d[p + '__loss_hist'], rtol=1e-5)
    assert _rel(s.eqn.sol_sampled_pts, d[p + '__sol']) < 1e-6


@pytest.mark.parametrize('structured', ['0', '1', '2'])
def test_burgers_through_solver_gp(solves, structured, monkeypatch):
    # (round 6: GPK_STRUCTURED=1 / 2 -- the structured solve and its Gram level exist for this system too; same fixtures of the reference's own run, same bounds)
    monkeypatch.setenv('GPK_STRUCTURED', structured)
    from src.solver import solver_GP
    d, p = solves, 'burgers_small'
    alpha, nu, st, sx, nug, steps, seed = d[p + '__params']
    cfg = argparse.Namespace(alpha=alpha, nu=nu, kernel='anisotropic_Gaussian', kernel_parameter=[st, sx], nugget=nug,
                             nugget_type='adaptive', GNsteps=int(steps), step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(int(seed))
    s = solver_GP(cfg, PDE_type='Burgers')
    s.set_equation(bdy=lambda x1, x2: -np.sin(np.pi * x2) * (x1 == 0) + 0 * (x2 == 0), rhs=lambda x1, x2: 0,
                   domain=np.array([[0, 1], [-1, 1]]), print_option=False)
    s.auto_sample(200, 60, print_option=False)
    assert s.eqn.N_boundary == 60 and np.array_equal(s.eqn.X_domain, d[p + '__X_domain'])
    s.solve(print_option=False)
    assert (s.eqn._prob.W1 is not None) == (structured != '0') and (s.eqn._prob.G is not None) == (structured == '2')
    np.testing.assert_allclose(s.eqn.loss_hist, d[p + '__loss_hist'], rtol=1e-6)
    np.testing.assert_allclose(s.eqn.ratio, d[p + '__ratio'], rtol=1e-13)
    assert _rel(s.eqn.sol_sampled_pts, d[p + '__sol']) < 1e-6
    s.test(d[p + '__X_test'], print_option=False)
    assert _rel(s.eqn.extended_sol, d[p + '__extended_sol']) < 1e-6
    assert s.eqn.Hessian_GN(s.eqn.init_sol).shape == (600, 600)


@pytest.mark.parametrize('structured', ['0', '1', '2'])
def test_eikonal_through_solver_gp(solves, structured, monkeypatch):
    # (round 6: GPK_STRUCTURED=1 / 2 -- the structured solve and its Gram level exist for this system too; same fixtures of the reference's own run, same bounds)
    monkeypatch.setenv('GPK_STRUCTURED', structured)
    from src.solver import solver_GP
    d, p = solves, 'eikonal_small'
    eps, sigma, nug, steps, seed = d[p + '__params']
    cfg = argparse.Namespace(eps=eps, kernel='Gaussian', kernel_parameter=sigma, nugget=nug, nugget_type='adaptive',
                             GNsteps=int(steps), step_size=1, initial_sol='zero', print_hist=False)
    np.random.seed(int(seed))
    s = solver_GP(cfg, PDE_type='Eikonal')
    s.set_equation(bdy=lambda x1, x2: 0, rhs=lambda x1, x2: 1, domain=np.array([[0, 1], [0, 1]]), print_option=False)
    s.auto_sample(200, 48, print_option=False)
    s.solve(print_option=False)
    np.testing.assert_allclose(s.eqn.loss_hist, d[p + '__loss_hist'], rtol=1e-6)
    assert _rel(s.eqn.sol_sampled_pts, d[p + '__sol']) < 1e-6
    s.test(d[p + '__X_test'], print_option=False)
    assert _rel(s.eqn.extended_sol, d[p + '__extended_sol']) < 1e-6


@pytest.mark.parametrize('structured', ['0', '1', '2'])
def test_darcy_through_solver_gp(solves, structured, monkeypatch):
    # (round 6: GPK_STRUCTURED=1 / 2 -- the structured solve and its Gram level exist for this system too; same fixtures of the reference's own run, same bounds)
    monkeypatch.setenv('GPK_STRUCTURED', structured)
    from src.solver import solver_GP
    d, p = solves, 'darcy_small'
    sigma, nug, steps, seed, ndata, noise = d[p + '__params']
    cfg = argparse.Namespace(kernel='Gaussian', kernel_parameter=sigma, nugget=nug, nugget_type='adaptive',
                             GNsteps=int(steps), step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(int(seed))
    s = solver_GP(cfg, PDE_type='Darcy_flow2d')
    s.set_equation(bdy=lambda x1, x2: 0, rhs=lambda x1, x2: 1, domain=np.array([[0, 1], [0, 1]]), print_option=False)
    s.auto_sample_IP(150, 40, int(ndata), print_option=False)
    s.get_observed_data(d[p + '__data_clean'], noise, print_option=False)      # noise drawn here, as in the reference run
    np.testing.assert_array_equal(s.eqn.data_u, d[p + '__data_u'])
    s.solve(print_option=False)
    np.testing.assert_allclose(s.eqn.loss_hist, d[p + '__loss_hist'], rtol=1e-6)
    assert _rel(s.eqn.sol_vec_a, d[p + '__sol_vec_a']) < 1e-6 and _rel(s.eqn.sol_vec_u, d[p + '__sol_vec_u']) < 1e-6
    s.test(d[p + '__X_test'], print_option=False)
    assert _rel(s.eqn.extended_sol_a, d[p + '__extended_sol_a']) < 1e-6
    assert _rel(s.eqn.extended_sol_u, d[p + '__extended_sol_u']) < 1e-6


def test_readme_command_line_runs():
    """BASELINE config 1: the reference README's command line against our driver; L2 errors of the reference run
    (2.24e-7 collocation, 2.56e-7 test at seed 0) are reproduced to the digit that conditioning allows (< 1e-6)."""
    env = dict(os.environ)
    code = ("import numpy, runpy, sys; numpy.random.seed(0); "
            "sys.argv=['main_NonLinElliptic2d.py','--kernel','Gaussian','--kernel_parameter','0.2','--nugget','1e-13',"
            "'--N_domain','900','--N_boundary','124','--GNsteps','4','--show_figure','']; "
            "runpy.run_path('main_NonLinElliptic2d.py', run_name='__main__')")
    r = subprocess.run([sys.executable, '-c', code], cwd=os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert '[Sample points] N_domain = 900, N_boundary = 124' in out
    l2 = [float(l.split('L2 error')[1]) for l in out.splitlines() if 'L2 error' in l]
    assert len(l2) == 2 and all(v < 1e-6 for v in l2), out[-1500:]


def test_other_drivers_run_small():
    pkg = os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')
    for args in (['main_Burgers1d.py', '--N_domain', '300', '--N_boundary', '90', '--GNsteps', '6', '--show_figure', ''],
                 ['main_Eikonal2d.py', '--N_domain', '300', '--N_boundary', '80', '--GNsteps', '6', '--show_figure', ''],
                 ['main_DarcyFlow2d.py', '--N_domain', '200', '--N_boundary', '60', '--N_data', '30', '--GNsteps', '6', '--show_figure', '']):
        r = subprocess.run([sys.executable] + args, cwd=pkg, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (args, r.stderr[-2000:])
        assert '[Gauss Newton] Gauss Newton iteration finished' in r.stdout and 'nan' not in r.stdout.lower(), r.stdout[-1500:]


def test_error_metrics_on_device_match_the_reference_definitions():
    """gpk_error_metrics against src/solver.py:175,191 of the reference: |truth - value|, its maximum, sqrt(sum of squares / n);
    odd lengths, a NaN (propagates like jnp.max), and through solver_GP.get_test_error with a scalar truth"""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(11)
    for n in (1, 63, 1024, 3601, 20001):
        t, a = rng.normal(size=n), rng.normal(size=n)
        err, mx, l2 = ctx.error_metrics(t, a)
        np.testing.assert_array_equal(err, np.abs(t - a))
        assert mx == np.max(np.abs(t - a))
        assert l2 == pytest.approx(np.sqrt(np.sum((t - a) ** 2) / n), rel=1e-14)
    t = rng.normal(size=100); a = t.copy(); a[37] = np.nan
    _, mx, l2 = ctx.error_metrics(t, a)
    assert np.isnan(mx) and np.isnan(l2)
    ctx.close()


def test_class_api_loss_history_is_exact(monkeypatch):
    """The loss history of GN_method is exact to rounding in EVERY entry.  Round 4 got there by calling gpk_gn_loss (true substitution) after
    every step, because the free in-step value of gpk_gn_step (F column of the GEMM-only solve, explicit diagonal-block inverses) is only good
    to ~1e-8 near convergence at nugget 1e-12; since round 5 gpk_gn_step itself reports the loss by true substitution (one vector, in front
    of the solve phase), so the default loop is one call per iteration.  Checked against the oracle's loss on the device's own factor for
    every iterate of the history; GPK_SEPARATE_LOSS=1 (round 4's sequence) gives the same iterates and the same numbers; gpk_tune(52, 0)
    restores the approximate in-step number (same iterates, losses within 1e-5)."""
    import gpk
    from oracle import gp_oracle as O
    from src.PDEs import Nonlinear_elliptic2d

    def run():
        np.random.seed(4)
        e = Nonlinear_elliptic2d(alpha=1.0, m=3, bdy=lambda x1, x2: O.elliptic_truth(x1, x2), rhs=lambda x1, x2: O.elliptic_rhs(x1, x2))
        e.sampled_pts(1200, 160)
        e.Gram_matrix(kernel='Gaussian', kernel_parameter=0.2, nugget=1e-12, nugget_type='adaptive')
        e.Gram_Cholesky()
        e.GN_method(max_iter=6, step_size=1, initial_sol='rdm', print_hist=False)
        return e

    e = run()
    assert len(e.loss_hist) == 7 and e.chol_info == 0
    sysm = O.EllipticSystem(1.0, 3.0, e.rhs_f, e.bdy_g)
    want_last = O.loss(sysm, [e.L], e.sol_sampled_pts)
    want_first = O.loss(sysm, [e.L], e.init_sol)
    assert e.loss_hist[-1] == pytest.approx(want_last, rel=1e-9)
    assert e.loss_hist[0] == pytest.approx(want_first, rel=1e-12)
    monkeypatch.setenv('GPK_SEPARATE_LOSS', '1')
    f = run()
    np.testing.assert_array_equal(f.sol_sampled_pts, e.sol_sampled_pts)       # same steps, only the way the numbers are obtained differs
    np.testing.assert_allclose(f.loss_hist, e.loss_hist, rtol=1e-9)            # both by true substitution (different launch shapes: not bitwise)
    assert f.loss_hist[-1] == e.loss_hist[-1]                                # the closing value always came from gpk_gn_loss
    monkeypatch.delenv('GPK_SEPARATE_LOSS')
    try:
        gpk.load_library().gpk_debug_set(52, 0)                               # the approximate number of rounds 2-4
        a = run()
    finally:
        gpk.load_library().gpk_debug_set(52, 1)
    np.testing.assert_array_equal(a.sol_sampled_pts, e.sol_sampled_pts)
    np.testing.assert_allclose(a.loss_hist, e.loss_hist, rtol=1e-5)


def test_step_with_the_concurrent_loss_chain_is_bitwise_repeatable():
    """Round 5: the exact loss of gpk_gn_step runs its substitution chain on the GEMM partition's stream NEXT TO the last panel chain of the
    pipelined phase and the backward solve of the tail (two single-vector solves at once, each with its own granule set).  Concurrency must
    not show in the numbers: the same five steps, 12 times over, give bit-identical loss histories and iterates (pipelined size: n_z = 1500),
    and the values are those of the serial placement (gpk_tune(52, 2): the chain in front of the solve)."""
    import gpk
    from oracle import gp_oracle as O
    ctx = gpk.Context(0)
    rng = np.random.RandomState(3)
    Nd, Nb = 1500, 200
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    z0 = rng.normal(size=Nd)
    T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive')
    assert ctx.potrf(T) == 0
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
    prob.workspace()

    def run():
        z = ctx.array(z0)
        hist = [ctx.gn_step(prob, z)[0] for _ in range(5)]
        sol = z.download(); z.free()
        return hist, sol
    ref = run()
    for _ in range(11):
        h, s = run()
        assert h == ref[0] and np.array_equal(s, ref[1])
    ctx.tune(52, 2)
    h, s = run()
    ctx.tune(52, 1)
    assert h == ref[0] and np.array_equal(s, ref[1])
    sysm = O.EllipticSystem(1.0, 3.0, f, g)
    L = np.tril(T.download())
    assert ref[0][0] == pytest.approx(O.loss(sysm, [L], z0), rel=1e-12)
    prob.release_workspace(); T.free(); ctx.close()

"""ORACLE parity at the sizes the metric is quoted on (round 4; the earlier full-size tests check properties only).

Every BASELINE configuration that fits one GPU, plus the north-star size, element by element against the CPU oracle on the
GPU box's host cores (numpy / scipy with the box's BLAS threads; seconds per comparison, the slow part is the oracle):

  * the WHOLE Gram matrix of the device against `O.gram_matrix_assembly` + `O.add_nugget`
    (reference src/Gram_matrice.py:11-187, src/PDEs.py:56-80,250-276, src/InverseProblems.py:66-103);
  * the device factor against numpy.linalg.cholesky (LAPACK dpotrf) of the SAME downloaded matrix (src/PDEs.py:75-80);
  * Gauss-Newton iterates and loss values against `O.gn_method` run on the device's factor, step by step
    (src/PDEs.py:104-135, 309-343; src/InverseProblems.py:153-186): iterate <= 1e-6 relative (the north star's bound),
    loss values rtol 1e-6.

Tolerances are written next to each assertion.  Where two backward-stable algorithms legitimately differ by eps * cond (the factor
of an ill-conditioned matrix) the bound is stated in those terms and the backward error is asserted beside it.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O

SIGMA = 0.2
UNIT = np.array([[0, 1], [0, 1]])


@pytest.fixture(scope='module')
def ctx():
    import gpk
    c = gpk.Context(0)
    yield c
    c.close()


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def _theta_check(name, got, want, diag_scale):
    """whole matrix: max |device - oracle| <= 4e-15 * (largest diagonal entry) -- the bound of the small-size fixtures
    (tests/test_gpu_parity.py); exactly symmetric"""
    assert got.shape == want.shape
    dev = float(np.max(np.abs(got - want)))
    print(f'\n[{name}] Theta order {got.shape[0]}: max |device - oracle| = {dev:.3e} (= {dev / diag_scale:.2e} of the largest diagonal entry)')
    assert dev <= 4e-15 * diag_scale
    assert np.array_equal(got, got.T)


def _steps_against_oracle(ctx, prob, sysm, Ls_host, z0, steps, name, loss_rtol=1e-6):
    """device gn_step vs O.gn_method on the same (device-computed, downloaded) factors, step by step"""
    z = ctx.array(z0)
    zo = np.array(z0, dtype=np.float64)
    worst = 0.0
    for it in range(steps):
        loss_dev, info = ctx.gn_step(prob, z, 1.0)                # loss of the iterate the step starts from
        assert info == 0
        t0 = time.perf_counter()
        zo_next, hist = O.gn_method(sysm, Ls_host, zo, 1, 1)      # triangular formulation (B2) of the reference step
        dt = time.perf_counter() - t0
        zd = z.download()
        r = _rel(zd, zo_next)
        worst = max(worst, r)
        print(f'[{name}] step {it + 1}: iterate rel. dev {r:.2e}, loss(start) device {loss_dev:.12e} oracle {hist[0]:.12e} '
              f'(rel {abs(loss_dev / hist[0] - 1):.1e}); oracle step {dt:.1f} s')
        assert r <= 1e-6                                          # north star: within 1e-6 relative of the reference path
        assert loss_dev == pytest.approx(hist[0], rel=loss_rtol)
        # follow the DEVICE trajectory: the next oracle step starts from the device's iterate, so every step is an independent
        # comparison of the same map (differences are not compounded through an ill-conditioned iteration)
        zo = zd.copy()
    final_dev = ctx.gn_loss(prob, z)
    final_or = O.loss(sysm, Ls_host, zo)
    assert final_dev == pytest.approx(final_or, rel=loss_rtol)
    z.free()
    return worst


def _structured_map_matches(ctx, prob, sprob, z0, steps, name, tol=1e-7):
    """Round 6: the optional structured solve (gpk_gn_structured_prepare: L^{-1}A(z) = W1 diag(d(z)) + W2) as a MAP, step by step along the
    default path's trajectory at FULL size: from the same iterate the structured step and the per-step solve must land within `tol` of each
    other (the per-step solve is within 1e-6 of the oracle at every one of these iterates: _steps_against_oracle above), same in-step loss."""
    z = ctx.array(z0)
    worst = 0.0
    for it in range(steps):
        zs = ctx.array(z.download())
        loss_d, info_d = ctx.gn_step(prob, z, 1.0)
        loss_s, info_s = ctx.gn_step(sprob, zs, 1.0)
        assert info_d == 0 and info_s == 0
        r = _rel(zs.download(), z.download())
        worst = max(worst, r)
        assert r <= tol, (name, it, r)
        assert loss_s == pytest.approx(loss_d, rel=1e-9)          # both by true substitution from the same iterate
        zs.free()
    print(f'[{name}] structured step vs per-step solve over {steps} steps of the trajectory: worst iterate rel. dev {worst:.2e}')
    z.free()
    return worst


# ------------------------------------------------------------------------------------------------ BASELINE config 2
@pytest.fixture(scope='module')
def c2(ctx):
    from src.sample_points import sampled_pts_rdm
    np.random.seed(0)                                             # bench.py::synthetic_problem
    Xd, Xb = sampled_pts_rdm(4000, 400, UNIT)
    z0 = np.random.normal(0.0, 1.0, 4000)
    return Xd, Xb, z0


def test_config2_whole_theta_and_factor_against_oracle(ctx, c2):
    Xd, Xb, _ = c2
    Nd, Nb = 4000, 400
    nug = 1e-9
    T, ratios = ctx.assemble('Nonlinear_elliptic', 'Gaussian', SIGMA, Xd, Xb, nug, 'adaptive')
    got = T.download()
    want, r = O.add_nugget(O.gram_matrix_assembly(Xd, Xb, 'Nonlinear_elliptic', 'Gaussian', SIGMA), 'Nonlinear_elliptic', Nd, Nb, nug)
    assert ratios[0] == pytest.approx(r[0], rel=1e-13)
    _theta_check('C2', got, want, 8.0 / SIGMA ** 4)
    # factor: device vs LAPACK on the SAME matrix
    assert ctx.potrf(T) == 0
    L = np.tril(T.download())
    t0 = time.perf_counter()
    Lref = np.linalg.cholesky(got)
    dt = time.perf_counter() - t0
    back_dev = np.linalg.norm(L @ L.T - got) / np.linalg.norm(got)
    back_ref = np.linalg.norm(Lref @ Lref.T - got) / np.linalg.norm(got)
    fro = _rel(L, Lref)
    # column-wise: where the two factors differ, in units of the column's own norm
    col = np.linalg.norm(L - Lref, axis=0) / np.linalg.norm(Lref, axis=0)
    dmin = float(np.min(np.diag(Lref)))
    print(f'[C2] factor order {L.shape[0]}: ||L_dev - L_lapack||_F / ||L||_F = {fro:.2e}; worst column {col.max():.2e} (column {int(col.argmax())}); '
          f'backward error device {back_dev:.2e} LAPACK {back_ref:.2e}; smallest pivot {dmin:.2e}; LAPACK {dt:.1f} s')
    assert back_dev <= 1e-14                                      # || L L^T - Theta || / || Theta ||: both are backward stable
    assert back_dev <= 16 * back_ref + 1e-15                      # (first GPU run: device 5.9e-16, LAPACK 8.1e-17 -- a few eps either way)
    # forward agreement of two backward-stable factorisations is bounded by eps * cond(Theta) per column at worst (measured: worst
    # column 2.1e-4 of its own norm, where the pivots are ~3e-5); the factor as a whole (Frobenius norm) must agree better than the
    # 1e-6 parity bound (measured 9.8e-9)
    assert fro <= 1e-6
    T.free()


def test_config2_gauss_newton_steps_against_oracle(ctx, c2):
    """the benchmark's own problem (nugget 1e-13, N(0,1) start, seed 0): all four steps of the reference configuration"""
    import gpk
    Xd, Xb, z0 = c2
    Nd, Nb = 4000, 400
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', SIGMA, Xd, Xb, 1e-13, 'adaptive')
    assert ctx.potrf(T) == 0
    L = np.tril(T.download())
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
    sysm = O.EllipticSystem(1.0, 3.0, f, g)
    # loss values at nugget 1e-13: ||L^{-1}F||^2 with cond(L) ~ 1e9 -- two exact-arithmetic-equivalent solves agree to ~1e-7 near
    # convergence (DESIGN section 4 'Numerics'); the ITERATE bound stays 1e-6
    _steps_against_oracle(ctx, prob, sysm, [L], z0, 4, 'C2', loss_rtol=1e-5)
    # and the reference's own operation sequence (B1: general LU solves of L, LU solve of H; src/PDEs.py:86,97,118) for the first step
    z = ctx.array(z0)
    ctx.gn_step(prob, z, 1.0)
    H, grad = O.gn_quantities(sysm, [L], z0, faithful=True)
    z1_b1 = z0 - np.linalg.solve(H, grad)
    r = _rel(z.download(), z1_b1)
    print(f'[C2] first step against the reference operation sequence (LU solves): iterate rel. dev {r:.2e}')
    assert r <= 1e-6
    prob.release_workspace(); T.free()


# ------------------------------------------------------------------------------------------------ BASELINE config 3
def test_config3_burgers_against_oracle(ctx):
    import gpk
    from src.sample_points import sampled_pts_rdm
    np.random.seed(0)                                             # main_Burgers1d.py:45 default seed
    dom = np.array([[0, 1], [-1, 1]])
    Xd, Xb = sampled_pts_rdm(2000, 400, dom, time_dependent=True)
    Nd, Nb = Xd.shape[0], Xb.shape[0]
    assert (Nd, Nb) == (2000, 399)
    kp, nug = [0.3, 0.05], 1e-5
    z0 = np.random.normal(0.0, 1.0, 3 * Nd)
    T, ratios = ctx.assemble('Burgers', 'anisotropic_Gaussian', kp, Xd, Xb, nug, 'adaptive')
    got = T.download()
    want, r = O.add_nugget(O.gram_matrix_assembly(Xd, Xb, 'Burgers', 'anisotropic_Gaussian', kp), 'Burgers', Nd, Nb, nug)
    np.testing.assert_allclose(ratios[:3], r, rtol=1e-13)
    _theta_check('C3', got, want, float(np.max(np.diag(want))))
    assert ctx.potrf(T) == 0
    L = np.tril(T.download())
    back = np.linalg.norm(L @ L.T - got) / np.linalg.norm(got)
    print(f'[C3] factor backward error {back:.2e}')
    assert back <= 1e-14
    bdy = -np.sin(np.pi * Xb[:, 1]) * (Xb[:, 0] == 0)
    prob = gpk.GNProblem(ctx, 'Burgers', Nd, Nb, np.zeros(Nd), bdy, T, p0=1.0, p1=0.02)
    sysm = O.BurgersSystem(1.0, 0.02, np.zeros(Nd), bdy)
    _steps_against_oracle(ctx, prob, sysm, [L], z0, 8, 'C3')           # all eight steps of the reference configuration
    sprob = gpk.GNProblem(ctx, 'Burgers', Nd, Nb, np.zeros(Nd), bdy, T, p0=1.0, p1=0.02, structured=True)
    _structured_map_matches(ctx, prob, sprob, z0, 8, 'C3')
    sprob.release_workspace(); prob.release_workspace(); T.free()


# ------------------------------------------------------------------------------------------------ BASELINE config 4
def test_config4_darcy_against_oracle(ctx):
    import gpk
    from src.sample_points import sampled_pts_rdm
    np.random.seed(9999)                                          # main_DarcyFlow2d.py:77 default seed
    Xd, Xb = sampled_pts_rdm(1600, 200, UNIT)
    Nd, Nb, Ndata, nug, noise = 1600, 200, 60, 1e-8, 1e-3
    z0 = np.random.normal(0.0, 1.0, 6 * Nd)
    data = 0.05 * np.sin(np.pi * Xd[:Ndata, 0]) * np.sin(np.pi * Xd[:Ndata, 1]) + noise * np.random.normal(0, 1.0, Ndata)
    Tu, ru = ctx.assemble('Darcy_u', 'Gaussian', SIGMA, Xd, Xb, nug, 'adaptive')
    Ta, ra = ctx.assemble('Darcy_a', 'Gaussian', SIGMA, Xd, Xb, nug, 'adaptive')
    wu, wa = O.gram_matrix_assembly(Xd, Xb, 'Darcy_flow2d', 'Gaussian', SIGMA)
    wu, r_u = O.add_nugget(wu, 'Darcy_u', Nd, Nb, nug)
    wa, r_a = O.add_nugget(wa, 'Darcy_a', Nd, Nb, nug)
    np.testing.assert_allclose(ru[:3], r_u, rtol=1e-13)
    np.testing.assert_allclose(ra[:2], r_a, rtol=1e-13)
    gu, ga = Tu.download(), Ta.download()
    _theta_check('C4 Theta_u', gu, wu, float(np.max(np.diag(wu))))
    _theta_check('C4 Theta_a', ga, wa, float(np.max(np.diag(wa))))
    assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
    Lu, La = np.tril(Tu.download()), np.tril(Ta.download())
    for nm, Lh, th in (('u', Lu, gu), ('a', La, ga)):
        back = np.linalg.norm(Lh @ Lh.T - th) / np.linalg.norm(th)
        print(f'[C4] factor {nm} backward error {back:.2e}')
        assert back <= 1e-14
    f = np.ones(Nd); g = np.zeros(Nb)
    prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=noise, data_u=data, L2=Ta)
    sysm = O.DarcySystem(f, g, data, noise)
    _steps_against_oracle(ctx, prob, sysm, [La, Lu], z0, 8, 'C4')     # all eight steps of the reference configuration
    assert prob.Wa is not None                                        # (the default: the iteration-independent a-part cached, bit-identical)
    sprob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=noise, data_u=data, L2=Ta, structured=True)
    _structured_map_matches(ctx, prob, sprob, z0, 8, 'C4')
    sprob.release_workspace(); prob.release_workspace(); Tu.free(); Ta.free()


# ------------------------------------------------------------------------------------------------ north-star size
def test_north_star_size_first_step_against_oracle(ctx):
    """N_domain = 10^4 (Theta of order 21000): the first two Gauss-Newton steps of the benchmark's problem against the oracle on the device's
    factor, and 24 random 256 x 256 blocks + the diagonal of Theta against the closed forms"""
    import gpk
    from src.sample_points import sampled_pts_rdm
    Nd, Nb = 10000, 1000
    N = 2 * Nd + Nb
    np.random.seed(0)
    Xd, Xb = sampled_pts_rdm(Nd, Nb, UNIT)
    z0 = np.random.normal(0.0, 1.0, Nd)
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    T = ctx.empty(N, N)
    nugget = 1e-13
    while True:
        _, ratios = ctx.assemble('Nonlinear_elliptic', 'Gaussian', SIGMA, Xd, Xb, nugget, 'adaptive', out=T)
        # ---- blocks of Theta (before the factorisation overwrites it)
        if nugget == 1e-13:
            rng = np.random.RandomState(5)
            Xdb = np.concatenate([Xd, Xb])
            worst = 0.0
            for b in range(24):
                r0, c0 = int(rng.randint(0, N - 256)), int(rng.randint(0, N - 256))
                if b < 6:                                         # six blocks on the diagonal (nugget), two of them across the seams
                    r0 = c0 = (Nd - 128, 2 * Nd - 128, r0, c0, 0, N - 256)[b]
                blk = T.download(rows=256, cols=256, row0=r0, col0=c0)
                rows, cols = np.arange(r0, r0 + 256), np.arange(c0, c0 + 256)
                want = np.empty((256, 256))
                for rs, rname in ((rows < Nd, 'lap'), (rows >= Nd, 'val')):
                    for cs, cname in ((cols < Nd, 'lap'), (cols >= Nd, 'val')):
                        if not rs.any() or not cs.any():
                            continue
                        X = Xd[rows[rs]] if rname == 'lap' else Xdb[rows[rs] - Nd]
                        Y = Xd[cols[cs]] if cname == 'lap' else Xdb[cols[cs] - Nd]
                        meth = {('lap', 'lap'): 'Delta_x_Delta_y_kappa', ('lap', 'val'): 'Delta_x_kappa',
                                ('val', 'lap'): 'Delta_y_kappa', ('val', 'val'): 'kappa'}[(rname, cname)]
                        want[np.ix_(rs, cs)] = O.deriv_kernel(meth, X[:, None, 0], X[:, None, 1], Y[None, :, 0], Y[None, :, 1], 'Gaussian', SIGMA)
                ii, jj = np.nonzero(rows[:, None] == cols[None, :])           # diagonal entries inside the block carry the nugget
                want[ii, jj] += np.where(rows[ii] < Nd, nugget * ratios[0], nugget)
                worst = max(worst, float(np.max(np.abs(blk - want))))
            print(f'\n[n10k] 24 random 256x256 blocks of Theta: max |device - oracle| = {worst:.3e}')
            assert worst <= 4e-15 * 8.0 / SIGMA ** 4
        info = ctx.potrf(T)
        if info == 0 or nugget > 1e-10:
            break
        nugget *= 10.0
    assert info == 0
    L = np.tril(T.download())
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
    sysm = O.EllipticSystem(1.0, 3.0, f, g)
    _steps_against_oracle(ctx, prob, sysm, [L], z0, 2, 'n10k', loss_rtol=1e-5)
    prob.release_workspace(); T.free()

"""Optional structured solve of the elliptic system (gpk_gn_structured_prepare): [L^{-1}A(z) | L^{-1}F(z)] formed from the precomputed
W1 = L^{-1}[I;0;0], W2 = L^{-1}[0;I;0], v0 = L^{-1}F(0) instead of a triangular solve per step.  Same iterates as the default
(reference operation sequence) path up to rounding, and as the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O


@pytest.mark.parametrize('Nd,Nb,nugget', [(300, 60, 1e-9), (1100, 160, 1e-10)])
def test_structured_step_matches_default_and_oracle(Nd, Nb, nugget):
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(Nd)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, nugget, 'adaptive')
    assert ctx.potrf(T) == 0
    z0 = rng.normal(size=Nd)
    out = []
    for structured in (False, True, 2):
        prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, structured=structured)
        z = ctx.array(z0)
        hist = []
        for _ in range(4):
            loss, info = ctx.gn_step(prob, z, 1.0)
            assert info == 0
            hist.append(loss)
        hist.append(ctx.gn_loss(prob, z))
        out.append((z.download().ravel().copy(), np.array(hist)))
    (za, ha), (zb, hb), (zc, hc) = out
    assert np.linalg.norm(zb - za) <= 1e-9 * np.linalg.norm(za)
    np.testing.assert_allclose(hb, ha, rtol=1e-7)
    assert np.linalg.norm(zc - za) <= 1e-8 * np.linalg.norm(za)          # Gram level: O(nz^2) assembly of the bordered matrix
    np.testing.assert_allclose(hc, ha, rtol=1e-6)
    if Nd <= 400:
        Theta = O.add_nugget(O.gram_matrix_assembly(Xd, Xb), 'Nonlinear_elliptic', Nd, Nb, nugget)[0]
        sol_ref, hist_ref = O.gn_method(O.EllipticSystem(1.0, 3.0, f, g), [O.cholesky(Theta)], z0, 4, 1)
        for zz in (za, zb, zc):                                          # both paths within the parity bound of the oracle (4 steps from a random start)
            assert np.linalg.norm(zz - sol_ref) <= 1e-6 * np.linalg.norm(sol_ref)
        np.testing.assert_allclose(hb, hist_ref, rtol=1e-6)
    ctx.lib.gpk_debug_set(40, 0)                                     # the switch: W1/W2/v0 present but ignored -> the default path, bit for bit
    try:
        z = ctx.array(z0)
        for _ in range(4):
            ctx.gn_step(prob, z, 1.0)
        assert np.array_equal(z.download().ravel(), za)
    finally:
        ctx.lib.gpk_debug_set(40, 1)
    ctx.close()


def _steps(ctx, prob, z0, n):
    z = ctx.array(z0)
    hist = []
    for _ in range(n):
        loss, info = ctx.gn_step(prob, z)
        assert info == 0
        hist.append(loss)
    hist.append(ctx.gn_loss(prob, z))
    out = z.download().copy()
    z.free()
    return out, np.array(hist)


def _run_structured_pair(system, make_prob, z0, nsteps, sysm, Ls, tol_pair, tol_hist=1e-6):
    """the per-step solve, the structured solve (W1 diag(d(z)) + W2, round 6) and the oracle on the same factor"""
    import gpk
    ctx, plain, structured = make_prob()
    assert structured.W1 is not None and structured.struct.W1 and plain.struct.W1 is None
    za, ha = _steps(ctx, plain, z0, nsteps)
    zb, hb = _steps(ctx, structured, z0, nsteps)
    assert np.linalg.norm(zb - za) <= tol_pair * np.linalg.norm(za), (system, np.linalg.norm(zb - za) / np.linalg.norm(za))
    np.testing.assert_allclose(hb, ha, rtol=tol_hist)
    sol_ref, hist_ref = O.gn_method(sysm, Ls, z0, nsteps, 1)
    assert np.linalg.norm(zb - sol_ref) <= 1e-6 * np.linalg.norm(sol_ref)           # the parity bound of the north star
    np.testing.assert_allclose(hb, hist_ref, rtol=1e-5)
    # the switch (gpk_tune key 40 = 0): W1 / W2 present but ignored -> the per-step solve, bit for bit
    ctx.tune(40, 0)
    zc, _ = _steps(ctx, structured, z0, nsteps)
    ctx.tune(40, 1)
    assert np.array_equal(zc, za)
    # second level (round 6 for these systems): the bordered matrix assembled from the Gram blocks of W1, W2 -- no solve of A(z), no product.
    # H is ill-conditioned here (1e10 .. 1e12): the assembled H agrees with S^T S to ~1e-14, the iterates to 1e-10 .. 1e-8
    structured.prepare_gram()
    assert structured.struct.G
    zg, hg = _steps(ctx, structured, z0, nsteps)
    assert np.linalg.norm(zg - za) <= max(100 * tol_pair, 1e-7) * np.linalg.norm(za), (system, 'gram', np.linalg.norm(zg - za) / np.linalg.norm(za))
    assert np.linalg.norm(zg - sol_ref) <= 1e-6 * np.linalg.norm(sol_ref)
    np.testing.assert_allclose(hg, hist_ref, rtol=1e-5)
    ctx.close()


def test_structured_solve_eikonal():
    """A(z) of the Eikonal system (reference src/PDEs.py:441-449) = A1 diag(d(z)) + A2, d = 2 v1 / eps, 2 v2 / eps, 0"""
    import gpk
    rng = np.random.RandomState(21)
    Nd, Nb, eps = 700, 120, 0.1
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = np.ones(Nd); g = np.zeros(Nb)
    hold = {}

    def make():
        ctx = gpk.Context(0)
        T, _ = ctx.assemble('Eikonal', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
        assert ctx.potrf(T) == 0
        hold['L'] = np.tril(T.download())
        return (ctx, gpk.GNProblem(ctx, 'Eikonal', Nd, Nb, f, g, T, p0=eps), gpk.GNProblem(ctx, 'Eikonal', Nd, Nb, f, g, T, p0=eps, structured=True))
    ctx, a, b = make()
    _run_structured_pair('Eikonal', lambda: (ctx, a, b), 0.1 * rng.normal(size=3 * Nd), 4, O.EikonalSystem(eps, f, g), [hold['L']], 1e-8)


def test_structured_solve_burgers():
    """A(z) of the Burgers system (reference src/PDEs.py:297-305): d = -alpha v2, -alpha v0, nu on the PDE row; slope-1/3 staircase"""
    import gpk
    rng = np.random.RandomState(22)
    Nd, Nb = 650, 99
    Xd = np.stack([rng.uniform(0, 1, Nd), rng.uniform(-1, 1, Nd)], axis=1)
    Xb = np.stack([rng.uniform(0, 1, Nb), rng.uniform(-1, 1, Nb)], axis=1)
    f = np.zeros(Nd); g = -np.sin(np.pi * Xb[:, 1]) * (rng.uniform(size=Nb) < 0.4)
    ctx = gpk.Context(0)
    T, _ = ctx.assemble('Burgers', 'anisotropic_Gaussian', [0.3, 0.05], Xd, Xb, 1e-5, 'adaptive')
    assert ctx.potrf(T) == 0
    L = np.tril(T.download())
    a = gpk.GNProblem(ctx, 'Burgers', Nd, Nb, f, g, T, p0=1.0, p1=0.02)
    b = gpk.GNProblem(ctx, 'Burgers', Nd, Nb, f, g, T, p0=1.0, p1=0.02, structured=True)
    _run_structured_pair('Burgers', lambda: (ctx, a, b), 0.2 * rng.normal(size=3 * Nd), 2, O.BurgersSystem(1.0, 0.02, f, g), [L], 1e-7)


@pytest.mark.parametrize('Nd,Nb,Ndata', [(333, 50, 20), (640, 128, 40)])
def test_structured_solve_darcy(Nd, Nb, Ndata):
    """A(z) of the Darcy system (reference src/InverseProblems.py:127-143): the only z-dependent entries sit in the v3 row of the u-part
    (f e^{-w0}, -v1, -v2, -w1, -w2); the a-part and the data rows are constants.  With and without the cached a-part."""
    import gpk
    rng = np.random.RandomState(23 + Nd)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = np.ones(Nd); g = np.zeros(Nb)
    data = 0.05 * np.sin(np.pi * Xd[:Ndata, 0]) * np.sin(np.pi * Xd[:Ndata, 1]) + 1e-3 * rng.normal(size=Ndata)
    for cache in (True, False):
        ctx = gpk.Context(0)
        Tu, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
        Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
        assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
        Lu, La = np.tril(Tu.download()), np.tril(Ta.download())
        a = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=1e-3, data_u=data, L2=Ta, cache_a=cache)
        b = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=1e-3, data_u=data, L2=Ta, cache_a=cache, structured=True)
        _run_structured_pair('Darcy', lambda: (ctx, a, b), 0.3 * np.random.RandomState(5).normal(size=6 * Nd), 3, O.DarcySystem(f, g, data, 1e-3), [La, Lu], 1e-7)


def test_structured_prepare_argument_checks():
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(1)
    Nd, Nb = 90, 30
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    T, _ = ctx.assemble('Eikonal', 'Gaussian', 0.2, Xd, Xb, 1e-8, 'adaptive')
    assert ctx.potrf(T) == 0
    with pytest.raises(Exception):                                # the other systems need the inverted diagonal blocks (GEMM-only solve path)
        gpk.GNProblem(ctx, 'Eikonal', Nd, Nb, np.ones(Nd), np.zeros(Nb), T, p0=0.1, structured=True, dinv=False)
    T2, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-8, 'adaptive')
    assert ctx.potrf(T2) == 0
    with pytest.raises(Exception):                                # no structured form of the relaxed system
        gpk.GNProblem(ctx, 'Nonlinear_elliptic_relaxed', Nd, Nb, np.ones(Nd), np.zeros(Nb), T2, p0=1.0, p1=3.0, pen_lambda=1e-4, structured=True)
    ctx.close()


@pytest.mark.parametrize('structured', [1, 2])
def test_in_step_loss_is_the_substituted_loss_in_the_structured_modes(structured):
    """Round 6 (advisor): the class API takes its loss history from gpk_gn_step's in-step loss.  In the structured / Gram modes that number
    used to be the mode's own approximate form; it is now the same true substitution gpk_gn_loss performs -- bit for bit -- at every
    iterate, also near convergence where the approximate forms kept three digits."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(77)
    Nd, Nb = 500, 90
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-12, 'adaptive')
    assert ctx.potrf(T) == 0
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, structured=structured)
    z = ctx.array(rng.normal(size=Nd))
    for _ in range(6):
        before = ctx.gn_loss(prob, z)
        loss, info = ctx.gn_step(prob, z)
        assert info == 0 and loss == before, (loss, before)
    ctx.close()

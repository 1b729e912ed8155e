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


def test_structured_prepare_rejects_other_systems():
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(1)
    Nd, Nb = 90, 30
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    T, _ = ctx.assemble('Eikonal', 'Gaussian', 0.2, Xd, Xb, 1e-8, 'adaptive')
    assert ctx.potrf(T) == 0
    with pytest.raises(Exception):
        gpk.GNProblem(ctx, 'Eikonal', Nd, Nb, np.ones(Nd), np.zeros(Nb), T, p0=0.1, structured=True)
    ctx.close()

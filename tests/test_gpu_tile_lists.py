"""Tile-list ("stream-K") GEMM launches (csrc/gpk_gemm.hip, GemmArgs::sk_segs): the host cuts a launch's slab iterations into one
share per resident workgroup slot; a tile cut between workgroups is finished by the last arriver in K order.  gpk_debug_set(42, 2)
forces the mode for every eligible launch, (42, 0) switches it off: same results as the one-tile-per-workgroup kernel within the
GEMM bound, bit-identical between runs (the order of arrival must not matter), for every operand layout, tile configuration and
launch kind of the library -- plain, lower-triangular output (SYRK), leading-zero operands, triangular operand (the inverted
diagonal blocks of the solve), the Cholesky factorisation's trailing updates -- and for the Gauss-Newton step as a whole."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O


@pytest.fixture(scope='module')
def ctx():
    import gpk
    c = gpk.Context(0)
    yield c
    c.lib.gpk_debug_set(42, 1); c.lib.gpk_debug_set(0, 0)
    c.close()


def _forced(ctx, cfg=0):
    ctx.lib.gpk_debug_set(42, 2); ctx.lib.gpk_debug_set(0, cfg)


def _auto(ctx):
    ctx.lib.gpk_debug_set(42, 1); ctx.lib.gpk_debug_set(0, 0)


@pytest.mark.parametrize('cfg', [0, 2, 3, 4])                        # automatic, 64x64, 128x64 (8 waves), 128x128 (16 waves)
@pytest.mark.parametrize('ta,tb,m,n,k,beta', [(0, 0, 700, 900, 1100, 0.0), (0, 1, 513, 640, 777, 1.0), (1, 0, 1000, 300, 2049, 0.0),
                                              (1, 1, 321, 1234, 515, 1.0), (0, 0, 64, 64, 4096, 1.0), (0, 1, 130, 70, 64, 0.0),
                                              (0, 0, 2500, 2500, 300, 1.0)])
def test_gemm_tile_lists(ctx, cfg, ta, tb, m, n, k, beta):
    rng = np.random.RandomState(m + n + k + cfg)
    A = rng.normal(size=(k, m) if ta else (m, k))
    B = rng.normal(size=(n, k) if tb else (k, n))
    Cm = rng.normal(size=(m, n))
    opA = A.T if ta else A
    opB = B.T if tb else B
    want = -1.0 * opA @ opB + beta * Cm
    dA, dB = ctx.array(A), ctx.array(B)
    bound = 1e-13 * (np.abs(opA) @ np.abs(opB) + np.abs(Cm)).max()
    _forced(ctx, cfg)
    try:
        runs = []
        for _ in range(3):
            dC = ctx.array(Cm)
            ctx.gemm(ta, tb, m, n, k, -1.0, dA, dB, beta, dC)
            runs.append(dC.download())
    finally:
        _auto(ctx)
    assert np.max(np.abs(runs[0] - want)) <= bound
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])


@pytest.mark.parametrize('n,k', [(700, 3000), (1500, 520), (65, 1000)])
def test_syrk_tile_lists(ctx, n, k):
    rng = np.random.RandomState(n + k)
    A = rng.normal(size=(k, n))
    C0 = rng.normal(size=(n, n))
    want = 0.5 * A.T @ A + 2.0 * C0
    dA = ctx.array(A)
    _forced(ctx)
    try:
        dC = ctx.array(C0)
        ctx.syrk(n, k, 0.5, dA, 2.0, dC)
        got = dC.download()
    finally:
        _auto(ctx)
    il = np.tril_indices(n)
    assert np.max(np.abs(got[il] - want[il])) <= 1e-13 * (np.abs(A.T) @ np.abs(A) + 2 * np.abs(C0)).max()
    ii, jj = np.triu_indices(n, 1)
    out = (ii // 64) < (jj // 64)                                     # (diagonal 64 x 64 tiles are written whole)
    assert np.array_equal(got[ii[out], jj[out]], C0[ii[out], jj[out]])   # tiles above the diagonal untouched


@pytest.mark.parametrize('n,nrhs,lead,block', [(1500, 901, 900, 256), (2304, 1200, 1199, 1024), (1000, 333, 0, 512)])
def test_solve_through_inverted_blocks_with_tile_lists(ctx, n, nrhs, lead, block):
    """gpk_trsm_dinv = triangular-operand products + leading-zero updates: every launch kind of the solve phase as a tile list"""
    rng = np.random.RandomState(n + nrhs)
    M = rng.normal(size=(n, n))
    L = np.linalg.cholesky(M @ M.T + n * np.eye(n))
    B = rng.normal(size=(n, nrhs))
    for c in range(min(nrhs, lead)):
        B[:max(0, lead - 1 - c), c] = 0.0
    from scipy.linalg import solve_triangular
    want = solve_triangular(L, B, lower=True)
    dL = ctx.array(L)
    D = ctx.trtri_diag(dL, block=block)
    outs = []
    for mode in (2, 0):                                               # forced tile lists, then never
        ctx.lib.gpk_debug_set(42, mode)
        try:
            dB = ctx.array(B)
            X = ctx.empty(n, nrhs); X.zero()
            ctx.trsm_dinv(dL, D, dB, X, lead=lead)
            outs.append(X.download())
        finally:
            _auto(ctx)
    for got in outs:
        assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)
    assert np.linalg.norm(outs[0] - outs[1]) <= 1e-13 * np.linalg.norm(want)


def test_cholesky_and_gn_step_with_tile_lists_everywhere(ctx):
    """the whole factor/solve path with tile lists forced on (including the pipelined product + factorisation of Hb, where they
    share the workspace with the split-K products of the same stream): iterates equal to the default schedule within rounding,
    and bitwise reproducible"""
    import gpk
    rng = np.random.RandomState(8)
    Nd, Nb = 1600, 200
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    z0 = rng.normal(size=Nd)
    sols = []
    for mode in (2, 2, 0):
        ctx.lib.gpk_debug_set(42, mode)
        try:
            T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-9, 'adaptive')
            assert ctx.potrf(T) == 0
            prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
            z = ctx.array(z0)
            for _ in range(3):
                loss, info = ctx.gn_step(prob, z)
                assert info == 0
            sols.append(z.download())
            prob.release_workspace(); T.free()
        finally:
            _auto(ctx)
    assert np.array_equal(sols[0], sols[1])
    assert np.linalg.norm(sols[0] - sols[2]) <= 1e-9 * np.linalg.norm(sols[2])


def test_lds_dma_feed_probe_kernel(dev_ctx):
    """the LDS-DMA experiment kernel (gpk_debug_gemm_dma, csrc/dev/gpk_gemm_dma_probe.hip; development build only): same bits as the
    product kernel -- same tile, same summation order -- on shapes with several K slabs and tiles"""
    ctx = dev_ctx
    rng = np.random.RandomState(3)
    for m, n, k in ((64, 64, 16), (192, 256, 208), (640, 128, 1024)):
        A = rng.normal(size=(m, k)); B = rng.normal(size=(k, n))
        dA, dB, dC, dR = ctx.array(A), ctx.array(B), ctx.empty(m, n), ctx.empty(m, n)
        assert ctx.lib.gpk_debug_gemm_dma(ctx.h, m, n, k, dA.ptr, dA.ld, dB.ptr, dB.ld, dC.ptr, dC.ld) == 0
        ctx.lib.gpk_debug_set(0, 2)
        try:
            ctx.gemm(0, 0, m, n, k, 1.0, dA, dB, 0.0, dR)
        finally:
            ctx.lib.gpk_debug_set(0, 0)
        got = dC.download()
        assert np.max(np.abs(got - A @ B)) <= 1e-13 * (np.abs(A) @ np.abs(B)).max()
        assert np.array_equal(got, dR.download())
    assert ctx.lib.gpk_debug_gemm_dma(ctx.h, 65, 64, 16, dA.ptr, dA.ld, dB.ptr, dB.ld, dC.ptr, dC.ld) < 0      # shape restrictions are checked

"""GPU parity tests: every entry point of the C ABI (include/gpk.h) against the CPU oracle on identical inputs.

Tolerances (fp64 everywhere):
  * Gram blocks: |GPU - oracle| <= 4e-15 * max|block|  (one exp + Hermite prefactors; ocml exp vs glibc exp <= 1 ulp)
  * GEMM/SYRK: 1e-13 relative to |A||B| row sums (summation order differs: MFMA k-chunks of 4)
  * POTRF / TRSM: backward-error style, ||L L^T - A|| <= 1e-13 ||A||, ||L X - B|| <= 1e-12 ||L|| ||X||
  * Gauss-Newton: solution vectors rel-L2 <= 1e-6 vs oracle / reference fixtures (the north-star bound);
    loss histories rtol 1e-6 at nugget >= 1e-8.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')
EQN = {'elliptic': 'Nonlinear_elliptic', 'burgers': 'Burgers', 'eikonal': 'Eikonal', 'darcy': 'Darcy_flow2d'}


@pytest.fixture(scope='module')
def ctx():
    import gpk
    c = gpk.Context(0)
    yield c
    c.close()


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ------------------------------------------------------------------------------------------------ assembly
def _layouts_for(eqn):
    return {'Nonlinear_elliptic': ['Nonlinear_elliptic'], 'Burgers': ['Burgers'], 'Eikonal': ['Eikonal'],
            'Darcy_flow2d': ['Darcy_u', 'Darcy_a']}[eqn]


def test_assemble_matches_reference_fixtures(ctx):
    d = np.load(os.path.join(G, 'theta_small.npz'))
    for name in sorted({k.split('__')[0] for k in d.files}):
        eqn = EQN[name.split('_')[0]]
        kp = d[name + '__kp']
        kernel, kp = ('Gaussian', float(kp[0])) if name.endswith('gauss') else ('anisotropic_Gaussian', [float(kp[0]), float(kp[1])])
        Xd, Xb, Xt = d[name + '__Xd'], d[name + '__Xb'], d[name + '__Xt']
        for lay in _layouts_for(eqn):
            key = {'Darcy_u': 'Theta_u', 'Darcy_a': 'Theta_a'}.get(lay, 'Theta')
            want = d[f'{name}__{key}']
            T, _ = ctx.assemble(lay, kernel, kp, Xd, Xb)
            got = T.download()
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= 4e-15 * np.max(np.abs(want)), (name, lay)
            assert np.array_equal(got, got.T), (name, lay, 'symmetry')
            want_t = d[f'{name}__{key}_test']
            got_t = ctx.assemble_test(lay, kernel, kp, Xt, Xd, Xb).download()
            assert np.max(np.abs(got_t - want_t)) <= 4e-15 * np.max(np.abs(want_t)), (name, lay, 'test')


@pytest.mark.parametrize('eqn,kernel,kp,Nd,Nb', [
    ('Nonlinear_elliptic', 'Gaussian', 0.2, 300, 44),
    ('Nonlinear_elliptic', 'Gaussian', 0.2, 257, 31),      # odd sizes: unaligned block offsets
    ('Burgers', 'anisotropic_Gaussian', [0.3, 0.05], 129, 33),
    ('Eikonal', 'Gaussian', 0.2, 200, 48),
    ('Darcy_flow2d', 'Gaussian', 0.2, 150, 40),
    ('Nonlinear_elliptic', 'Gaussian', 0.2, 1, 0),         # degenerate: single point, no boundary
])
def test_assemble_vs_oracle_with_nugget(ctx, eqn, kernel, kp, Nd, Nb):
    rng = np.random.RandomState(Nd)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    T = O.gram_matrix_assembly(Xd, Xb, eqn, kernel, kp)
    Ts = T if isinstance(T, tuple) else (T,)
    for lay, Tref in zip(_layouts_for(eqn), Ts):
        for ntype in ('adaptive', 'identity', 'none'):
            want, ratios = O.add_nugget(Tref, lay, Nd, Nb, 1e-3, ntype)
            Td, r = ctx.assemble(lay, kernel, kp, Xd, Xb, 1e-3, ntype)
            got = Td.download()
            assert np.max(np.abs(got - want)) <= 4e-15 * np.max(np.abs(want)), (lay, ntype)
            if ntype == 'adaptive' and ratios:
                np.testing.assert_allclose(r[:len(ratios)], ratios, rtol=1e-14)


def test_elliptic_trace_ratio_kat(ctx):
    """reference notebook KAT: 900/124 points, sigma 0.2 -> 4394.531249999999 (nugget independent)."""
    rng = np.random.RandomState(0)
    _, r = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, rng.uniform(0, 1, (900, 2)), rng.uniform(0, 1, (124, 2)), 1e-8, 'adaptive')
    assert r[0] == pytest.approx(4394.531249999999, rel=1e-14)


def test_extend_fused_matches_matrix_product(ctx):
    rng = np.random.RandomState(3)
    Xd = rng.uniform(0, 1, (211, 2)); Xb = rng.uniform(0, 1, (37, 2)); Xt = rng.uniform(0, 1, (301, 2))
    for eqn, lay, kernel, kp in [('Nonlinear_elliptic', 'Nonlinear_elliptic', 'Gaussian', 0.2),
                                 ('Burgers', 'Burgers', 'anisotropic_Gaussian', [0.3, 0.1]),
                                 ('Eikonal', 'Eikonal', 'Gaussian', 0.25), ('Darcy_flow2d', 'Darcy_a', 'Gaussian', 0.2)]:
        Tt = O.construct_theta_test(Xt, Xd, Xb, eqn, kernel, kp)
        Tt = Tt[1] if lay == 'Darcy_a' else (Tt[0] if isinstance(Tt, tuple) else Tt)
        coeff = rng.normal(size=Tt.shape[1])
        got = ctx.extend(lay, kernel, kp, Xt, Xd, Xb, coeff).download()
        want = Tt @ coeff
        assert np.max(np.abs(got - want)) <= 1e-12 * (np.abs(Tt) @ np.abs(coeff)).max(), lay


# ------------------------------------------------------------------------------------------------ dense
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('m,n,k', [(64, 64, 16), (130, 70, 33), (257, 513, 100), (1, 7, 5), (1500, 2700, 129), (300, 300, 1)])
def test_gemm(ctx, ta, tb, m, n, k):
    rng = np.random.RandomState(m + n + k)
    A = rng.normal(size=(k, m) if ta else (m, k))          # asymmetric operands catch transposed maps
    B = rng.normal(size=(n, k) if tb else (k, n))
    Cm = rng.normal(size=(m, n))
    opA = A.T if ta else A
    opB = B.T if tb else B
    want = 1.7 * opA @ opB - 0.3 * Cm
    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
    ctx.gemm(ta, tb, m, n, k, 1.7, dA, dB, -0.3, dC)
    got = dC.download()
    bound = 1e-13 * (np.abs(opA) @ np.abs(opB) + np.abs(Cm)).max()
    assert np.max(np.abs(got - want)) <= bound


@pytest.mark.parametrize('cfg', [1, 2, 3, 4])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('m,n,k', [(130, 70, 33), (257, 513, 100), (1, 7, 5), (700, 900, 129), (128, 64, 16)])
def test_gemm_forced_tile_configurations(ctx, cfg, ta, tb, m, n, k):
    """gpk_debug_set(0, c): 128x128 tiles (4 waves), 64x64 tiles, the 128x64 tile with 8 waves and the 128x128 tile with 16 waves (large launches) on
    ragged shapes -- every instantiation of the kernel template is exercised whatever the size heuristics pick"""
    rng = np.random.RandomState(m + n + k + cfg)
    A = rng.normal(size=(k, m) if ta else (m, k))
    B = rng.normal(size=(n, k) if tb else (k, n))
    Cm = rng.normal(size=(m, n))
    opA = A.T if ta else A
    opB = B.T if tb else B
    want = 1.7 * opA @ opB - 0.3 * Cm
    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
    ctx.lib.gpk_debug_set(0, cfg)
    try:
        ctx.gemm(ta, tb, m, n, k, 1.7, dA, dB, -0.3, dC)
        got = dC.download()
    finally:
        ctx.lib.gpk_debug_set(0, 0)
    bound = 1e-13 * (np.abs(opA) @ np.abs(opB) + np.abs(Cm)).max()
    assert np.max(np.abs(got - want)) <= bound


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('m,n,k,alpha', [(1000, 64, 448, -1.0), (300, 33, 200, 1.0), (257, 64, 65, -1.0), (4000, 50, 512, -1.0), (640, 64, 64, -1.0)])
def test_gemm_update_kernels(ctx, ta, tb, m, n, k, alpha):
    """C <- C +- A B with a tall-and-skinny C (ragged K and N): the rank-<=64 one-shot kernel and the shapes of the left-looking
    updates of the Cholesky chain"""
    rng = np.random.RandomState(m + n + k)
    A = rng.normal(size=(k, m) if ta else (m, k))
    B = rng.normal(size=(n, k) if tb else (k, n))
    Cm = rng.normal(size=(m, n))
    opA = A.T if ta else A
    opB = B.T if tb else B
    want = alpha * opA @ opB + Cm
    dA, dB, dC = ctx.array(A), ctx.array(B), ctx.array(Cm)
    ctx.gemm(ta, tb, m, n, k, alpha, dA, dB, 1.0, dC)
    got = dC.download()
    bound = 1e-13 * (np.abs(opA) @ np.abs(opB) + np.abs(Cm)).max()
    assert np.max(np.abs(got - want)) <= bound


@pytest.mark.parametrize('split', [2, 3, 7])
@pytest.mark.parametrize('ta,tb,m,n,k,beta', [(1, 0, 700, 130, 2100, 0.0), (0, 1, 500, 64, 1500, 1.0), (0, 0, 333, 200, 1111, 1.0), (1, 1, 96, 96, 4000, 0.0)])
def test_gemm_split_k(ctx, split, ta, tb, m, n, k, beta):
    """gpk_debug_set(25, s): every tile's K range cut into s chunks on separate workgroups, combined by the last arriver in
    chunk order -- same bound as the unsplit kernel, and bit-identical between runs (the order of arrival does not matter)"""
    rng = np.random.RandomState(m + n + k + split)
    A = rng.normal(size=(k, m) if ta else (m, k))
    B = rng.normal(size=(n, k) if tb else (k, n))
    Cm = rng.normal(size=(m, n))
    opA = A.T if ta else A
    opB = B.T if tb else B
    want = -1.0 * opA @ opB + beta * Cm
    dA, dB = ctx.array(A), ctx.array(B)
    ctx.lib.gpk_debug_set(25, split)
    try:
        runs = []
        for _ in range(3):
            dC = ctx.array(Cm)
            ctx.gemm(ta, tb, m, n, k, -1.0, dA, dB, beta, dC)
            runs.append(dC.download())
    finally:
        ctx.lib.gpk_debug_set(25, 0)
    bound = 1e-13 * (np.abs(opA) @ np.abs(opB) + np.abs(Cm)).max()
    assert np.max(np.abs(runs[0] - want)) <= bound
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    dC = ctx.array(Cm)                                               # and the unsplit launch still works on the same handle
    ctx.gemm(ta, tb, m, n, k, -1.0, dA, dB, beta, dC)
    assert np.max(np.abs(dC.download() - want)) <= bound


def test_gemm_unaligned_leading_dimension(ctx):
    """odd ld / odd offsets take the scalar-load path"""
    rng = np.random.RandomState(5)
    m, n, k = 150, 90, 77
    A = rng.normal(size=(m, k)); B = rng.normal(size=(k, n))
    dA = ctx.empty(m, k, ld=k + 1).upload(A); dB = ctx.empty(k, n, ld=n + 3).upload(B); dC = ctx.empty(m, n, ld=n + 1)
    ctx.gemm(0, 0, m, n, k, 1.0, dA, dB, 0.0, dC)
    assert np.max(np.abs(dC.download() - A @ B)) <= 1e-13 * (np.abs(A) @ np.abs(B)).max()


@pytest.mark.parametrize('n,k', [(100, 300), (1000, 517), (2600, 64)])
def test_syrk_lower_and_full(ctx, n, k):
    rng = np.random.RandomState(n)
    A = rng.normal(size=(k, n))
    want = A.T @ A
    dA = ctx.array(A); dC = ctx.empty(n, n); dC.zero()
    ctx.syrk(n, k, 1.0, dA, 0.0, dC, full=True)
    got = dC.download()
    assert np.max(np.abs(got - want)) <= 1e-13 * (np.abs(A.T) @ np.abs(A)).max()
    assert np.array_equal(got, got.T)


def _spd(rng, n):
    M = rng.normal(size=(n, n))
    return M @ M.T + n * np.eye(n)


@pytest.mark.parametrize('n', [1, 5, 63, 64, 65, 128, 200, 333, 1000, 2049])
def test_potrf(ctx, n):
    rng = np.random.RandomState(n)
    A = _spd(rng, n)
    dA = ctx.array(A)
    info = ctx.potrf(dA)
    assert info == 0
    ctx.tril(dA)
    L = dA.download()
    assert np.allclose(np.triu(L, 1), 0.0)
    assert np.linalg.norm(L @ L.T - A) <= 1e-13 * np.linalg.norm(A)
    Lref = np.linalg.cholesky(A)
    assert np.linalg.norm(L - Lref) <= 1e-11 * np.linalg.norm(Lref)


@pytest.mark.parametrize('panel', [1, 3])
def test_potrf_grid_larger_than_resident(dev_ctx, panel):
    """Panel kernels with more workgroups than the chip holds at once (the third design keeps one workgroup per CU: 261 > 256): the
    workgroup that stores the diagonal block waits for the others' load tickets, so it must be one that is dispatched AFTER them (a
    wait for a later workgroup starves when that workgroup is bound to the waiting one's CU).  Checked through sampled entries of L L^T."""
    ctx = dev_ctx
    n, k = 16640, 48
    rng = np.random.RandomState(3)
    M = rng.normal(size=(k, n))
    dM = ctx.array(M)
    dA = ctx.empty(n, n)
    ctx.syrk(n, k, 1.0, dM, 0.0, dA, full=True)
    A = dA.download()
    A[np.arange(n), np.arange(n)] += n
    dA.upload(A)
    ctx.lib.gpk_debug_set(21, 1 if panel == 3 else panel)            # 3: second design, rolled instantiation
    ctx.lib.gpk_debug_set(41, 0 if panel == 3 else 1)
    try:
        info = ctx.potrf(dA)
    finally:
        ctx.lib.gpk_debug_set(21, 1); ctx.lib.gpk_debug_set(41, 1)
    assert info == 0
    L = np.tril(dA.download())
    ii = rng.randint(0, n, 64); jj = rng.randint(0, n, 64)
    for i, j in zip(ii, jj):
        assert abs(L[i] @ L[j] - A[i, j]) <= 1e-12 * n
    rows = rng.randint(0, n, 4)
    assert np.max(np.abs(L[rows] @ L.T - A[rows])) <= 1e-12 * n
    dA.free(); dM.free()


def test_potrf_reports_first_bad_pivot(ctx):
    rng = np.random.RandomState(1)
    n = 300
    A = _spd(rng, n)
    A[150, 150] = -1.0                                       # leading 150x150 minor is SPD, pivot 151 fails
    dA = ctx.array(A)
    info = ctx.potrf(dA)
    assert info == 151
    L = dA.download()
    assert np.isnan(np.tril(L)[151:, 150:]).any()            # NaNs propagate like jnp.linalg.cholesky


@pytest.mark.parametrize('trans', [0, 1])
@pytest.mark.parametrize('n,nrhs', [(64, 64), (65, 3), (200, 130), (1000, 257), (1537, 1000), (300, 1)])
def test_trsm(ctx, trans, n, nrhs):
    rng = np.random.RandomState(n + nrhs)
    L = np.tril(rng.normal(size=(n, n))) + np.diag(rng.uniform(3, 4, n) * np.sqrt(n))
    B = rng.normal(size=(n, nrhs))
    dL, dB = ctx.array(L), ctx.array(B)
    ctx.trsm(dL, dB, trans=bool(trans))
    X = dB.download().reshape(n, nrhs)
    op = L.T if trans else L
    assert np.linalg.norm(op @ X - B) <= 1e-12 * np.linalg.norm(L) * np.linalg.norm(X)


@pytest.mark.parametrize('n,nrhs,lead', [(64, 64, 0), (256, 100, 0), (300, 77, 0), (1000, 257, 0), (1537, 1001, 0), (2360, 1101, 1100),
                                         (1924, 901, 900), (700, 650, 649), (513, 300, 200), (4300, 513, 500)])
@pytest.mark.parametrize('block', [256, 512, 1024, 2048])
def test_trsm_dinv(ctx, n, nrhs, lead, block):
    """All-GEMM forward solve through the explicit inverses of the diagonal blocks (256 .. 2048 rows; gpk_trtri_diag + gpk_trsm_dinv)
    against numpy and against the substitution path; lead > 0: right-hand sides with the leading-zero shape of [A | F]."""
    rng = np.random.RandomState(n + nrhs)
    L = np.tril(rng.normal(size=(n, n))) + np.diag(rng.uniform(3, 4, n) * np.sqrt(n))
    B = rng.normal(size=(n, nrhs))
    if lead:
        rows = np.arange(n)[:, None]; cols = np.arange(nrhs)[None, :]
        B[(cols < lead) & (rows < lead - 1 - cols)] = 0.0
    dL = ctx.array(L)
    D = ctx.trtri_diag(dL, block=block)
    Dh = D.download()
    for k0 in range(0, n, block):                                 # each block: inverse of the diagonal block, exact zeros above
        nk = min(block, n - k0)
        blk = Dh[k0:k0 + nk, :nk]
        assert np.all(np.triu(blk, 1) == 0.0)
        assert np.linalg.norm(L[k0:k0 + nk, k0:k0 + nk] @ blk - np.eye(nk)) <= 1e-12 * nk
    dB = ctx.array(B)
    dX = ctx.empty(n, nrhs); dX.zero()
    ctx.trsm_dinv(dL, D, dB, dX, lead=lead)
    X = dX.download().reshape(n, nrhs)
    assert np.linalg.norm(L @ X - B) <= 1e-12 * np.linalg.norm(L) * np.linalg.norm(X)
    dB2 = ctx.array(B)
    ctx.trsm(dL, dB2)
    X2 = dB2.download().reshape(n, nrhs)
    assert np.linalg.norm(X - X2) <= 1e-12 * np.linalg.norm(X2)
    if lead:                                                      # structural zeros of the solution stay exact zeros
        assert np.all(X[(cols < lead) & (rows < lead - 1 - cols)] == 0.0)
    with pytest.raises(Exception):
        ctx.trsm_dinv(dL, D, dB, dB)                              # aliasing is refused


@pytest.mark.parametrize('mode', [1, 0])                          # data-tagged hand-offs (default), two launches per block (the flag-chained form was removed in round 6)
@pytest.mark.parametrize('n', [1, 64, 100, 1000, 3001])
def test_trsv_and_potrs(dev_ctx, n, mode):
    ctx = dev_ctx
    rng = np.random.RandomState(n)
    A = _spd(rng, n)
    b = rng.normal(size=n)
    dA = ctx.array(A)
    assert ctx.potrf(dA) == 0
    ctx.lib.gpk_debug_set(4, mode)
    try:
        xs = []
        for _ in range(3):                                        # repeated: the tags / flags of earlier solves must not satisfy a later one
            db = ctx.array(b)                                     # contiguous vector -> single-vector path
            ctx.potrs(dA, db)
            xs.append(db.download())
    finally:
        ctx.lib.gpk_debug_set(4, 1)
    x = xs[0]
    assert np.linalg.norm(A @ x - b) <= 1e-11 * np.linalg.norm(A) * np.linalg.norm(x)
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0], xs[2])


# ------------------------------------------------------------------------------------------------ Gauss-Newton
def _factor_on_device(ctx, layout, kernel, kp, Xd, Xb, nugget):
    T, ratios = ctx.assemble(layout, kernel, kp, Xd, Xb, nugget, 'adaptive')
    info = ctx.potrf(T)
    assert info == 0
    return T, ratios


def _gn_run(ctx, prob, init, steps, step_size=1.0):
    z = ctx.array(init)
    hist = []
    for _ in range(steps):
        loss, info = ctx.gn_step(prob, z, step_size)
        assert info == 0
        hist.append(loss)
    hist.append(ctx.gn_loss(prob, z))
    return z.download(), np.array(hist)


def test_gn_elliptic_fixture(ctx):
    import gpk
    d = np.load(os.path.join(G, 'solves.npz')); p = 'elliptic_small'
    alpha, m, sigma, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor_on_device(ctx, 'Nonlinear_elliptic', 'Gaussian', sigma, Xd, Xb, nug)
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Xd.shape[0], Xb.shape[0], d[p + '__rhs_f'], d[p + '__bdy_g'], L, p0=alpha, p1=m)
    sol, hist = _gn_run(ctx, prob, d[p + '__init_sol'], int(steps))
    np.testing.assert_allclose(hist, d[p + '__loss_hist'], rtol=1e-5)
    assert _rel(sol, d[p + '__sol']) < 1e-6
    # measurement vector and fused extension
    sv = ctx.gn_measurement(prob, ctx.array(sol))
    assert _rel(sv, d[p + '__sol_vec']) < 1e-6
    coeff = ctx.array(sv)
    ctx.potrs(L, coeff)
    ext = ctx.extend('Nonlinear_elliptic', 'Gaussian', sigma, d[p + '__X_test'], Xd, Xb, coeff).download()
    assert _rel(ext, d[p + '__extended_sol']) < 1e-6


def test_gn_elliptic_baseline_config1(ctx):
    """BASELINE config 1 on the reference's own points/initial guess: solution within 1e-6 of the reference run."""
    import gpk
    d = np.load(os.path.join(G, 'solves.npz')); p = 'elliptic_c1'
    alpha, m, sigma, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor_on_device(ctx, 'Nonlinear_elliptic', 'Gaussian', sigma, Xd, Xb, nug)
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', 900, 124, d[p + '__rhs_f'], d[p + '__bdy_g'], L, p0=alpha, p1=m)
    sol, hist = _gn_run(ctx, prob, d[p + '__init_sol'], int(steps))
    assert _rel(sol, d[p + '__sol']) < 1e-6
    truth = O.elliptic_truth(Xd[:, 0], Xd[:, 1])
    assert np.sqrt(np.sum((truth - sol) ** 2) / 900) < 1e-6


def test_gn_hessian_and_gradient_api(ctx):
    import gpk
    rng = np.random.RandomState(4)
    Nd, Nb = 120, 28
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    T = O.gram_matrix_assembly(Xd, Xb, 'Nonlinear_elliptic', 'Gaussian', 0.2)
    Lr = O.cholesky(O.add_nugget(T, 'Nonlinear_elliptic', Nd, Nb, 1e-6)[0])
    L, _ = _factor_on_device(ctx, 'Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-6)
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, L, p0=1.0, p1=3.0)
    z = rng.normal(size=Nd)
    H, grad = ctx.gn_hessian_grad(prob, ctx.array(z))
    Hr, gr = O.gn_quantities(O.EllipticSystem(1.0, 3.0, f, g), [Lr], z)
    assert _rel(H, Hr) < 1e-7 and _rel(grad, gr) < 1e-7
    assert np.array_equal(H, H.T)


def test_gn_burgers_fixture(ctx):
    import gpk
    d = np.load(os.path.join(G, 'solves.npz')); p = 'burgers_small'
    alpha, nu, st, sx, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, ratios = _factor_on_device(ctx, 'Burgers', 'anisotropic_Gaussian', [st, sx], Xd, Xb, nug)
    np.testing.assert_allclose(ratios, d[p + '__ratio'], rtol=1e-13)
    prob = gpk.GNProblem(ctx, 'Burgers', Xd.shape[0], Xb.shape[0], d[p + '__rhs_f'], d[p + '__bdy_g'], L, p0=alpha, p1=nu)
    z, hist = _gn_run(ctx, prob, d[p + '__init_sol'], int(steps))
    np.testing.assert_allclose(hist, d[p + '__loss_hist'], rtol=1e-6)
    assert _rel(z[:Xd.shape[0]], d[p + '__sol']) < 1e-6
    assert _rel(ctx.gn_measurement(prob, ctx.array(z)), d[p + '__sol_vec']) < 1e-6


def test_gn_eikonal_fixture(ctx):
    import gpk
    d = np.load(os.path.join(G, 'solves.npz')); p = 'eikonal_small'
    eps, sigma, nug, steps, _ = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor_on_device(ctx, 'Eikonal', 'Gaussian', sigma, Xd, Xb, nug)
    prob = gpk.GNProblem(ctx, 'Eikonal', Xd.shape[0], Xb.shape[0], d[p + '__rhs_f'], d[p + '__bdy_g'], L, p0=eps)
    z, hist = _gn_run(ctx, prob, d[p + '__init_sol'], int(steps))
    np.testing.assert_allclose(hist, d[p + '__loss_hist'], rtol=1e-6)
    assert _rel(z[:Xd.shape[0]], d[p + '__sol']) < 1e-6


def test_gn_darcy_fixture(ctx):
    import gpk
    d = np.load(os.path.join(G, 'solves.npz')); p = 'darcy_small'
    sigma, nug, steps, _, ndata, noise = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    Nd, Nb = Xd.shape[0], Xb.shape[0]
    Lu, _ = _factor_on_device(ctx, 'Darcy_u', 'Gaussian', sigma, Xd, Xb, nug)
    La, _ = _factor_on_device(ctx, 'Darcy_a', 'Gaussian', sigma, Xd, Xb, nug)
    prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, d[p + '__rhs_f'], d[p + '__bdy_g'], Lu, p0=noise,
                         data_u=d[p + '__data_u'], L2=La)
    z, hist = _gn_run(ctx, prob, d[p + '__init_sol'], int(steps))
    np.testing.assert_allclose(hist, d[p + '__loss_hist'], rtol=1e-6)
    F = ctx.gn_measurement(prob, ctx.array(z))
    assert _rel(F[:3 * Nd], d[p + '__sol_vec_a']) < 1e-6
    assert _rel(F[3 * Nd:7 * Nd + Nb], d[p + '__sol_vec_u']) < 1e-6


def test_gn_relaxed_fixture(ctx):
    import gpk
    d = np.load(os.path.join(G, 'solves.npz')); p = 'elliptic_relaxed'
    alpha, m, sigma, nug, steps, _, lam = d[p + '__params']
    Xd, Xb = d[p + '__X_domain'], d[p + '__X_boundary']
    L, _ = _factor_on_device(ctx, 'Nonlinear_elliptic', 'Gaussian', sigma, Xd, Xb, nug)
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic_relaxed', Xd.shape[0], Xb.shape[0], d[p + '__rhs_f'], d[p + '__bdy_g'], L,
                         p0=alpha, p1=m, pen_lambda=lam)
    z, hist = _gn_run(ctx, prob, d[p + '__init_sol'], int(steps))
    np.testing.assert_allclose(hist, d[p + '__loss_hist'], rtol=1e-5)
    assert _rel(z[Xd.shape[0]:], d[p + '__sol']) < 1e-6


def test_notebook_kat_on_device(ctx):
    """Elliptic notebook known-answer vector (reference-committed numbers) reproduced by the HIP path."""
    import gpk
    import nb_recipes as R
    np.random.seed(10)
    Xd, Xb = R.notebook_sample_points(900, 124)
    L, ratios = _factor_on_device(ctx, 'Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-4)
    assert ratios[0] == pytest.approx(R.ELLIPTIC_RATIO, rel=1e-14)
    init = np.random.normal(0.0, 1.0, 900)
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', 900, 124, O.elliptic_rhs(Xd[:, 0], Xd[:, 1]), O.elliptic_truth(Xb[:, 0], Xb[:, 1]), L, p0=1.0, p1=3.0)
    sol, hist = _gn_run(ctx, prob, init, 5)
    np.testing.assert_allclose(hist, R.ELLIPTIC_J, rtol=1e-8)
    err = np.abs(O.elliptic_truth(Xd[:, 0], Xd[:, 1]) - sol)
    assert np.sqrt(np.sum(err ** 2) / 900) == pytest.approx(R.ELLIPTIC_PTS_L2, rel=1e-7)


def test_microbenchmarks_run(dev_ctx):
    ctx = dev_ctx
    tf = ctx.ubench_mfma_f64(5000)
    bw = ctx.ubench_hbm_write(1 << 28, 5)
    print(f'\n[ubench] v_mfma_f64_16x16x4_f64: {tf:.1f} TFLOP/s   streaming fp64 stores: {bw:.0f} GB/s')
    assert tf > 1.0 and bw > 100.0


@pytest.mark.parametrize('layout,kernel,kp', [('Nonlinear_elliptic', 'Gaussian', 0.2), ('Burgers', 'anisotropic_Gaussian', [0.3, 0.05]),
                                              ('Eikonal', 'Gaussian', 0.15), ('Darcy_a', 'Gaussian', 0.2)])
def test_assembly_two_points_per_lane_is_bit_identical(ctx, layout, kernel, kp):
    """gpk_assemble picks the kernel with two column points per lane (16-byte stores) when every block offset and size is even
    (BASELINE configs 2, 4, 5); gpk_debug_set(47, 0) forces the one-point-per-lane kernel: same bits, and the oracle's values."""
    rng = np.random.RandomState(17)
    Nd, Nb = 310, 62
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    got = []
    for pairs in (1, 0):
        ctx.lib.gpk_debug_set(47, pairs)
        try:
            T, _ = ctx.assemble(layout, kernel, kp, Xd, Xb, 1e-6, 'adaptive')
            got.append(T.download())
        finally:
            ctx.lib.gpk_debug_set(47, 1)
    assert np.array_equal(got[0], got[1])
    eqn = {'Darcy_a': 'Darcy_flow2d'}.get(layout, layout)
    ref = O.gram_matrix_assembly(Xd, Xb, eqn, kernel, kp)
    ref = ref[1] if layout == 'Darcy_a' else ref
    want, _ = O.add_nugget(ref, layout, Nd, Nb, 1e-6)
    assert np.max(np.abs(got[0] - want)) <= 4e-15 * np.max(np.abs(want))

"""Schedule variants must agree.  Every fused / reordered kernel added for speed keeps its plainer predecessor behind
gpk_debug_set (development aid, include/gpk.h): the same Gauss-Newton steps are run under each variant at a size where
all of them are active (256-row strips, several Cholesky panels, several chained TRSV blocks, > 1 supertile) and the
iterates compared -- same operation, different schedule, so agreement is to rounding (amplified by the conditioning of
Theta at the nugget used)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O

VARIANTS = [
    ('default', {}),
    ('64-row base solves instead of 256-row strips', {3: 0}),
    ('two launches per TRSV block', {4: 0}),
    ('supertile schedule of the leading-zero SYRK', {6: 1}),
    ('128x128 GEMM tiles', {0: 1}),
    ('128x64 tiles with 8 waves for every launch', {0: 3}),
    ('64x64 tiles also for the large launches', {33: 0, 38: 0}),
    ('128x128 tiles with 16 waves for every launch', {0: 4}),
    ('leading-zero launches in bands of 1 MB of A (every launch banded)', {35: 1, 36: 1}),
    ('no banded tile orders', {35: 0, 36: 0}),
    ('substitution strips instead of the inverted diagonal blocks', {10: 0}),
    ('substitution, 64-row base solves', {10: 0, 3: 0}),
    ('product then right-looking factorisation on one stream (no CU-partition pipeline)', {12: 0}),
    ('right-looking rank-64 updates inside the block columns of the pipelined factorisation', {18: 0}),
    ('pipeline with two pre-fork blocks and a 64-CU chain partition', {17: 2, 13: 64}),
    ('leading-zero products walking K downwards', {16: 1}),
    ('second-design panel kernel in its rolled instantiation', {41: 0}),
    ('Cholesky of Theta on the two-partition pipeline as well', {20: 100000}),
    ('pipelined products without split-K', {24: 0}),
    ('pipelined products with 32-row tiles / with 128-row tiles', {34: 0}),
    ('pipelined products with 128-row tiles', {34: 128}),
    ('pipelined products split into many K chunks', {24: 4000}),
    ('block update of the pipeline in three parts (look-ahead over the last panel of the previous block)', {26: 1}),
    ('look-ahead, narrow first block, 256-column blocks', {26: 1, 28: 128, 29: 256}),
    ('192-column blocks, right-looking chain, look-ahead', {26: 1, 28: 64, 29: 192, 18: 0}),
]
DEFAULTS = {0: 0, 3: 1, 4: 1, 5: 1, 6: 0, 7: 0, 10: 1, 12: 1, 13: 32, 14: 7000, 16: 0, 17: 1, 18: 1, 20: 0, 21: 1, 24: 1000, 26: 0, 28: 512, 29: 512, 33: 1500, 34: 64, 35: 192, 36: 256, 38: 6000, 41: 1}


def _run(ctx, variant, Xd, Xb, f, g, init, steps, nugget):
    import gpk
    for k, v in DEFAULTS.items():
        ctx.lib.gpk_debug_set(k, v)
    for k, v in variant.items():
        ctx.lib.gpk_debug_set(k, v)
    try:
        T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, nugget, 'adaptive')
        assert ctx.potrf(T) == 0
        prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Xd.shape[0], Xb.shape[0], f, g, T, p0=1.0, p1=3.0)
        z = ctx.array(init)
        hist = []
        for _ in range(steps):
            loss, info = ctx.gn_step(prob, z, 1.0)
            assert info == 0
            hist.append(loss)
        hist.append(ctx.gn_loss(prob, z))
        return z.download().ravel().copy(), np.array(hist)
    finally:
        for k, v in DEFAULTS.items():
            ctx.lib.gpk_debug_set(k, v)


def test_schedule_variants_agree():
    # (the superseded designs of rounds 1-2 -- keys 5 = 0, 7 = 1, 21 = 0 / 2, 4 = 2 -- were removed in round 6; both builds reject them)
    import gpk
    ctx = gpk.Context(0, dev=True)
    np.random.seed(5)
    from src.sample_points import sampled_pts_rdm
    Nd, Nb = 1100, 160                                           # N = 2360 (10 strips), n_z = 1100 (18 panels / TRSV blocks, 3 pipeline blocks)
    Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    init = np.random.normal(0.0, 1.0, Nd)
    for k, v in ((5, 0), (7, 1), (21, 0), (21, 2), (4, 2)):
        assert ctx.lib.gpk_tune(ctx.h, k, v) < 0
    ref_z, ref_hist = _run(ctx, {}, Xd, Xb, f, g, init, 3, 1e-9)
    truth = O.elliptic_truth(Xd[:, 0], Xd[:, 1])
    assert np.sqrt(np.mean((ref_z - truth) ** 2)) < 1e-4
    for name, variant in VARIANTS[1:]:
        z, hist = _run(ctx, variant, Xd, Xb, f, g, init, 3, 1e-9)
        assert np.max(np.abs(z - ref_z)) <= 1e-7 * np.max(np.abs(ref_z)), name
        np.testing.assert_allclose(hist, ref_hist, rtol=1e-5, err_msg=name)
    ctx.close()


def test_latency_probes_run():
    import gpk
    ctx = gpk.Context(0, dev=True)
    vals = [ctx.ubench_latency(m) for m in range(7)]
    print('\n[ubench] cycles/op: dep fma64 %.1f, indep fma64 %.1f, dep ds_read %.1f, indep ds_read %.1f, dep mfma %.1f, '
          'dep global (1 line) %.1f, dep global (16 lines) %.1f' % tuple(vals))
    assert all(0.5 < v < 5000 for v in vals)
    ctx.close()


def test_eikonal_leading_zero_layout_agrees_with_dense_schedule():
    """Eikonal system: gn_step regroups the unknowns (v1, v2, v0) so that A(z) has the staircase shape the elliptic system has and
    skips the structural zeros; gpk_debug_set(23, 0) runs the dense schedule.  Same operation -> same iterates, and both agree with
    the oracle."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(11)
    Nd, Nb, eps = 700, 120, 0.1                                  # N = 2920, n_z = 2100: several inverted blocks, 5 pipeline blocks
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = np.ones(Nd); g = np.zeros(Nb)
    T, _ = ctx.assemble('Eikonal', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
    assert ctx.potrf(T) == 0
    L = np.tril(T.download())
    z0 = np.zeros(3 * Nd)
    out = {}
    try:
        for mode in (1, 2, 0):                                    # exact two-segment profile (default), conservative closed form, dense
            ctx.lib.gpk_debug_set(23, mode)
            prob = gpk.GNProblem(ctx, 'Eikonal', Nd, Nb, f, g, T, p0=eps)
            z = ctx.array(z0)
            hist = []
            for _ in range(4):
                loss, info = ctx.gn_step(prob, z)
                assert info == 0
                hist.append(loss)
            hist.append(ctx.gn_loss(prob, z))
            out[mode] = (z.download().copy(), np.array(hist))
    finally:
        ctx.lib.gpk_debug_set(23, 1)
    for m in (1, 2):
        assert np.max(np.abs(out[m][0] - out[0][0])) <= 1e-9 * np.max(np.abs(out[0][0])), m
        np.testing.assert_allclose(out[m][1], out[0][1], rtol=1e-8)
    sol_ref, hist_ref = O.gn_method(O.EikonalSystem(eps, f, g), [L], z0, 4, 1)
    assert np.linalg.norm(out[1][0] - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref)
    np.testing.assert_allclose(out[1][1], hist_ref, rtol=1e-6)
    ctx.close()


def test_burgers_leading_zero_layout_agrees_with_dense_schedule():
    """Burgers system: the three unknowns of a collocation point all enter A(z) first in row t; gn_step interleaves them (3t + group),
    which gives a staircase of slope 1/3, and skips the zeros above it; gpk_debug_set(23, 0) runs the dense schedule."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(12)
    Nd, Nb = 650, 99                                             # N = 2699, n_z = 1950 (ragged everywhere)
    Xd = np.stack([rng.uniform(0, 1, Nd), rng.uniform(-1, 1, Nd)], axis=1)
    Xb = np.stack([rng.uniform(0, 1, Nb), rng.uniform(-1, 1, Nb)], axis=1)
    f = np.zeros(Nd); g = -np.sin(np.pi * Xb[:, 1]) * (rng.uniform(size=Nb) < 0.4)
    T, _ = ctx.assemble('Burgers', 'anisotropic_Gaussian', [0.3, 0.05], Xd, Xb, 1e-5, 'adaptive')
    assert ctx.potrf(T) == 0
    L = np.tril(T.download())
    z0 = rng.normal(size=3 * Nd)
    out = {}
    try:
        for mode in (1, 0):
            ctx.lib.gpk_debug_set(23, mode)
            prob = gpk.GNProblem(ctx, 'Burgers', Nd, Nb, f, g, T, p0=1.0, p1=0.02)
            z = ctx.array(z0)
            hist = []
            for _ in range(2):                                   # (from a random start the iteration amplifies rounding differences quickly)
                loss, info = ctx.gn_step(prob, z)
                assert info == 0
                hist.append(loss)
            hist.append(ctx.gn_loss(prob, z))
            out[mode] = (z.download().copy(), np.array(hist))
    finally:
        ctx.lib.gpk_debug_set(23, 1)
    assert np.linalg.norm(out[1][0] - out[0][0]) <= 1e-8 * np.linalg.norm(out[0][0])
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-6)   # (this start diverges: losses grow, rounding differences with them)
    sol_ref, hist_ref = O.gn_method(O.BurgersSystem(1.0, 0.02, f, g), [L], z0, 2, 1)
    assert np.linalg.norm(out[1][0] - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref)
    np.testing.assert_allclose(out[1][1], hist_ref, rtol=1e-6)
    ctx.close()


@pytest.mark.parametrize('Nd,Nb,Ndata', [(333, 50, 20), (640, 128, 40), (1100, 150, 60)])
def test_darcy_leading_zero_layout_agrees_with_dense_schedule(Nd, Nb, Ndata):
    """Darcy system (round 4): gn_step orders the unknowns v1, v2, w1, w2, w0, v0; the u-part of A(z) then has a piecewise staircase
    (slope 1, a flat step, slope 1 again -- GpkStair with three segments), the a-part a slope-1 staircase on a contiguous sub-range of
    columns; the solve and the products skip the zeros above them.  gpk_debug_set(23, 0) runs the dense schedule.  Sizes that are not
    multiples of anything (333: no column boundary of the profile is tile-aligned), exactly aligned (640) and several inverted blocks
    (1100: factors of order 4550 / 3300).  Same operation -> same iterates; both against the oracle."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(Nd)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = np.ones(Nd); g = np.zeros(Nb)
    data = 0.05 * np.sin(np.pi * Xd[:Ndata, 0]) * np.sin(np.pi * Xd[:Ndata, 1]) + 1e-3 * rng.normal(size=Ndata)
    Tu, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
    Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
    assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
    Lu, La = np.tril(Tu.download()), np.tril(Ta.download())
    z0 = 0.3 * rng.normal(size=6 * Nd)
    out = {}
    try:
        for mode in (1, 0):
            ctx.lib.gpk_debug_set(23, mode)
            prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=1e-3, data_u=data, L2=Ta)
            z = ctx.array(z0)
            hist = []
            for _ in range(3):
                loss, info = ctx.gn_step(prob, z)
                assert info == 0
                hist.append(loss)
            hist.append(ctx.gn_loss(prob, z))
            out[mode] = (z.download().copy(), np.array(hist))
            prob.release_workspace()
    finally:
        ctx.lib.gpk_debug_set(23, 1)
    assert np.linalg.norm(out[1][0] - out[0][0]) <= 1e-8 * np.linalg.norm(out[0][0])
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-7)
    sysm = O.DarcySystem(f, g, data, 1e-3)
    sol_ref, hist_ref = O.gn_method(sysm, [La, Lu], z0, 3, 1)
    assert np.linalg.norm(out[1][0] - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref)
    np.testing.assert_allclose(out[1][1], hist_ref, rtol=1e-6)
    ctx.close()


def test_darcy_steps_are_bitwise_reproducible():
    """The Darcy step in the leading-zero layout (piecewise profile in the solve, two product launches, one-stream factorisation of the
    order-6601 H): the same three steps from the same start, five times -- the bits must not depend on how the workgroups arrive."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(31)
    Nd, Nb, Ndata = 1100, 150, 60
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    data = 0.05 * rng.normal(size=Ndata)
    Tu, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-7, 'adaptive')
    Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-7, 'adaptive')
    assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
    prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, np.ones(Nd), np.zeros(Nb), Tu, p0=1e-3, data_u=data, L2=Ta)
    z0 = 0.3 * rng.normal(size=6 * Nd)
    ref = None
    for _ in range(5):
        z = ctx.array(z0)
        losses = []
        for _ in range(3):
            loss, info = ctx.gn_step(prob, z)
            assert info == 0
            losses.append(loss)
        out = z.download().copy()
        z.free()
        if ref is None:
            ref = (out, losses)
        else:
            assert np.array_equal(out, ref[0]) and losses == ref[1]
    ctx.close()


@pytest.mark.parametrize('n', [64, 65, 130, 1000, 2049, 4001])
def test_fused_panel_schedule_matches_separate_update_launches(n):
    """gpk_debug_set(48, .): the rank-64 work between two Cholesky panels rides inside the panel kernels (default: part B at the
    start of the panel workgroups, part A in extra workgroups next to the factorisation) or runs as launches of its own (round 2).
    Same factor up to rounding, same LAPACK info; the Gauss-Newton step (left-looking chain of the pipelined phase) likewise."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(n)
    M = rng.normal(size=(n, n))
    A0 = M @ M.T + n * np.eye(n)
    out = []
    try:
        for fused in (1, 0):
            ctx.lib.gpk_debug_set(48, fused)
            A = ctx.array(A0)
            assert ctx.potrf(A) == 0
            out.append(np.tril(A.download()))
            A.free()
        Lref = np.linalg.cholesky(A0)
        for L in out:
            assert np.linalg.norm(L - Lref) <= 1e-12 * np.linalg.norm(A0)
        assert np.linalg.norm(out[0] - out[1]) <= 1e-13 * np.linalg.norm(A0)
        bad = A0.copy()
        k = (2 * n) // 3
        bad[k, k] = -1.0
        ctx.lib.gpk_debug_set(48, 1)
        A = ctx.array(bad)
        assert ctx.potrf(A) == k + 1
    finally:
        ctx.lib.gpk_debug_set(48, 1)
        ctx.close()


def test_fused_panel_schedule_in_the_pipelined_gn_step():
    import gpk
    from oracle import gp_oracle as O
    ctx = gpk.Context(0)
    rng = np.random.RandomState(5)
    Nd, Nb = 1700, 200                                            # n_z + 1 = 1701: pipelined product + factorisation (3 blocks and more)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    z0 = rng.normal(size=Nd)
    sols = []
    try:
        for fused in (2, 0):                                      # 2: also the left-looking chain of the pipelined phase (off by default there)
            ctx.lib.gpk_debug_set(48, fused)
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
        ctx.lib.gpk_debug_set(48, 1)
        ctx.close()
    assert np.linalg.norm(sols[0] - sols[1]) <= 1e-9 * np.linalg.norm(sols[1])


@pytest.mark.parametrize('Nd,Nb,Ndata', [(333, 50, 20), (640, 128, 40), (1100, 150, 60)])
def test_darcy_cached_a_part_is_bit_identical_to_the_per_step_path(Nd, Nb, Ndata):
    """Round 6: the a-part rows of the Darcy system ([w1; w2; w0] against L_a, reference src/InverseProblems.py:137-146) do not involve
    z_old -- W_a = L_a^{-1} A_a and W_a^T W_a are the same in every step.  gpk_gn_darcy_prepare computes them once with the step's own
    launches (default of GNProblem; cache_a=False / GPK_DARCY_CACHE=0 recompute them every step): iterates, losses, the step vector and
    the pivot status must agree BIT FOR BIT, at sizes where no boundary of the profile is tile-aligned, where all are, and with several
    inverted blocks per factor; the cached path against the oracle as well."""
    import gpk
    ctx = gpk.Context(0)
    rng = np.random.RandomState(7 + Nd)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = np.ones(Nd); g = np.zeros(Nb)
    data = 0.05 * np.sin(np.pi * Xd[:Ndata, 0]) * np.sin(np.pi * Xd[:Ndata, 1]) + 1e-3 * rng.normal(size=Ndata)
    Tu, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
    Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
    assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
    Lu, La = np.tril(Tu.download()), np.tril(Ta.download())
    z0 = 0.3 * rng.normal(size=6 * Nd)
    out = {}
    for cached in (True, False):
        prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=1e-3, data_u=data, L2=Ta, cache_a=cached)
        assert (prob.Wa is not None) == cached and bool(prob.struct.Ha) == cached
        z = ctx.array(z0)
        hist, deltas = [], []
        for _ in range(3):
            loss, info = ctx.gn_step(prob, z)
            assert info == 0
            hist.append(loss)
            deltas.append(prob.workspace()[2].download().copy())
        hist.append(ctx.gn_loss(prob, z))
        out[cached] = (z.download().copy(), hist, deltas)
        prob.release_workspace()
    assert np.array_equal(out[True][0], out[False][0]) and out[True][1] == out[False][1]
    for a, b in zip(out[True][2], out[False][2]):
        assert np.array_equal(a, b)
    sol_ref, hist_ref = O.gn_method(O.DarcySystem(f, g, data, 1e-3), [La, Lu], z0, 3, 1)
    assert np.linalg.norm(out[True][0] - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref)
    np.testing.assert_allclose(out[True][1], hist_ref, rtol=1e-6)
    # bad arguments of the prepare call: wrong system, missing inverted blocks, leading dimensions too small
    prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, f, g, Tu, p0=1e-3, data_u=data, L2=Ta, cache_a=False)
    S = prob.workspace()[0]
    import ctypes as C
    W = gpk.DeviceArray(ctx, 3 * Nd, 3 * Nd, gpk.device.pad_ld(3 * Nd))
    assert ctx.lib.gpk_gn_darcy_prepare(ctx.h, C.byref(prob.struct), S.ptr, S.ld, W.ptr, 3 * Nd - 1, W.ptr, W.ld) < 0
    st = gpk.device.GNProblemStruct.from_buffer_copy(prob.struct)
    st.Dinv2 = None
    assert ctx.lib.gpk_gn_darcy_prepare(ctx.h, C.byref(st), S.ptr, S.ld, W.ptr, W.ld, W.ptr, W.ld) < 0
    ctx.close()

"""Empty and degenerate inputs through the C ABI: zero-sized operands are no-ops (no launch with an empty grid, no fault),
an assembly without boundary points works, argument errors come back as error codes with a message, not as crashes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O


@pytest.fixture(scope='module')
def ctx():
    import gpk
    c = gpk.Context(0)
    yield c
    c.close()


def test_zero_sized_operands_are_noops(ctx):
    A = ctx.array(np.eye(4))
    assert ctx.lib.gpk_potrf(ctx.h, A.ptr, 0, A.ld, None) == 0
    B = ctx.array(np.ones((4, 3)))
    assert ctx.lib.gpk_trsm(ctx.h, 0, A.ptr, 4, A.ld, B.ptr, 0, B.ld) == 0          # no right-hand sides
    assert ctx.lib.gpk_trsm(ctx.h, 0, A.ptr, 0, A.ld, B.ptr, 3, B.ld) == 0          # empty system
    assert ctx.lib.gpk_gemm(ctx.h, 0, 0, 0, 3, 4, 1.0, A.ptr, A.ld, B.ptr, B.ld, 0.0, B.ptr, B.ld) == 0
    assert ctx.lib.gpk_gemm(ctx.h, 0, 0, 4, 0, 4, 1.0, A.ptr, A.ld, B.ptr, B.ld, 0.0, B.ptr, B.ld) == 0
    ctx.synchronize()
    np.testing.assert_array_equal(B.download(), np.ones((4, 3)))
    # K = 0: C <- beta * C
    Cm = ctx.array(np.full((4, 3), 2.0))
    assert ctx.lib.gpk_gemm(ctx.h, 0, 0, 4, 3, 0, 1.0, A.ptr, A.ld, B.ptr, B.ld, 0.5, Cm.ptr, Cm.ld) == 0
    ctx.synchronize()
    np.testing.assert_allclose(Cm.download(), np.full((4, 3), 1.0))


def test_assembly_without_boundary_points(ctx):
    rng = np.random.RandomState(0)
    Xd = rng.uniform(0, 1, (37, 2)); Xb = np.zeros((0, 2))
    T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.3, Xd, Xb, 0.0, 'none')
    want = O.gram_matrix_assembly(Xd, Xb, 'Nonlinear_elliptic', 'Gaussian', 0.3)
    got = T.download()
    assert got.shape == want.shape == (74, 74)
    assert np.max(np.abs(got - want)) <= 4e-15 * np.max(np.abs(want))


def test_argument_errors_are_reported(ctx):
    import gpk
    A = ctx.array(np.eye(4))
    rc = ctx.lib.gpk_potrf(ctx.h, A.ptr, 8, 4, None)                                 # lda < n
    assert rc != 0
    with pytest.raises(gpk.GpkError):
        ctx._chk(rc)
    assert ctx.lib.gpk_trsm(ctx.h, 0, None, 4, 4, A.ptr, 1, 1) != 0                 # null factor


def test_inverted_block_entry_points_degenerate_and_errors(ctx):
    """gpk_trtri_diag / gpk_trsm_dinv: empty systems are no-ops, a bad block size or aliased operands are refused"""
    A = ctx.array(np.eye(4) * 2.0)
    D = ctx.empty(4, 256, ld=256)
    B = ctx.array(np.ones((4, 3))); X = ctx.empty(4, 3); X.zero()
    assert ctx.lib.gpk_trtri_diag(ctx.h, A.ptr, 0, A.ld, D.ptr, 256) == 0
    assert ctx.lib.gpk_trsm_dinv(ctx.h, A.ptr, D.ptr, 256, 0, A.ld, B.ptr, 3, B.ld, X.ptr, X.ld, 0) == 0
    assert ctx.lib.gpk_trsm_dinv(ctx.h, A.ptr, D.ptr, 256, 4, A.ld, B.ptr, 0, B.ld, X.ptr, X.ld, 0) == 0
    assert ctx.lib.gpk_trtri_diag(ctx.h, A.ptr, 4, A.ld, D.ptr, 100) != 0           # block size not 256 / 512 / 1024
    assert ctx.lib.gpk_trsm_dinv(ctx.h, A.ptr, D.ptr, 300, 4, A.ld, B.ptr, 3, B.ld, X.ptr, X.ld, 0) != 0
    assert ctx.lib.gpk_trsm_dinv(ctx.h, A.ptr, D.ptr, 256, 4, A.ld, B.ptr, 3, B.ld, B.ptr, B.ld, 0) != 0      # X aliases B
    assert ctx.lib.gpk_trsm_dinv(ctx.h, A.ptr, None, 256, 4, A.ld, B.ptr, 3, B.ld, X.ptr, X.ld, 0) != 0       # null inverses
    # and the tiny valid case: diag(2) -> X = B / 2
    assert ctx.lib.gpk_trtri_diag(ctx.h, A.ptr, 4, A.ld, D.ptr, 256) == 0
    assert ctx.lib.gpk_trsm_dinv(ctx.h, A.ptr, D.ptr, 256, 4, A.ld, B.ptr, 3, B.ld, X.ptr, X.ld, 0) == 0
    ctx.synchronize()
    np.testing.assert_allclose(X.download(), np.full((4, 3), 0.5))


def test_one_context_alternating_problem_shapes():
    """The Gauss-Newton step keeps its out-of-place solve buffer in the handle and relies on the part it never writes being zero:
    steps of problems with different shapes, layouts (reversed / plain) and systems through ONE context must give what each
    problem gives in a context of its own."""
    import gpk

    def setup(ctx, Nd, Nb, seed):
        rng = np.random.RandomState(seed)
        Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
        f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
        T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-8, 'adaptive')
        assert ctx.potrf(T) == 0
        prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, dinv=256)
        return prob, ctx.array(rng.normal(size=Nd))

    shapes = [(700, 120, 1), (1100, 90, 2), (700, 120, 1), (640, 64, 3)]
    want = []
    for Nd, Nb, seed in shapes[:2] + shapes[3:]:
        c = gpk.Context(0)
        prob, z = setup(c, Nd, Nb, seed)
        for _ in range(2):
            c.gn_step(prob, z)
        H, grad = c.gn_hessian_grad(prob, z)                      # plain layout through the same workspace
        c.gn_step(prob, z)
        want.append((z.download().copy(), H.copy()))
        c.close()
    want = [want[0], want[1], want[0], want[2]]
    c = gpk.Context(0)
    probs = [setup(c, Nd, Nb, seed) for Nd, Nb, seed in shapes]
    for _ in range(2):                                            # interleave the steps of the four problems
        for prob, z in probs:
            c.gn_step(prob, z)
    for (prob, z), (zw, Hw) in zip(probs, want):
        H, grad = c.gn_hessian_grad(prob, z)
        c.gn_step(prob, z)
        np.testing.assert_array_equal(H, Hw)
        np.testing.assert_array_equal(z.download(), zw)
    c.close()


def test_round4_entry_points_reject_bad_arguments_and_handle_edges():
    """gpk_tune (unknown key, development-only variants in the product library), gpk_error_metrics (n = 1, n <= 0, null pointers),
    gpk_prof_read_assembly before any timed assembly, Darcy in the leading-zero layout without data points and without boundary
    points."""
    import ctypes as C
    import gpk
    from oracle import gp_oracle as O
    ctx = gpk.Context(0)
    assert ctx.lib.gpk_tune(ctx.h, 9999, 1) < 0
    for key, value in ((5, 0), (7, 1), (21, 0), (21, 2), (4, 2)):               # superseded designs of rounds 1-2: removed in round 6
        assert ctx.lib.gpk_tune(ctx.h, key, value) < 0, (key, value)
        assert 'removed in round 6' in ctx.lib.gpk_last_error(ctx.h).decode()
    for key, value in ((11, 32), (54, 1)):                                      # probes / measured-and-not-adopted: development build only
        assert ctx.lib.gpk_tune(ctx.h, key, value) < 0, (key, value)
        assert 'development build' in ctx.lib.gpk_last_error(ctx.h).decode()
    assert ctx.lib.gpk_tune(ctx.h, 21, 1) == 0 and ctx.lib.gpk_tune(ctx.h, 4, 0) == 0 and ctx.lib.gpk_tune(ctx.h, 4, 1) == 0
    dev = gpk.Context(0, dev=True)
    assert dev.lib.gpk_tune(dev.h, 54, 1) == 0 and dev.lib.gpk_tune(dev.h, 54, 0) == 0      # ... and it has that one
    assert dev.lib.gpk_tune(dev.h, 21, 2) < 0                                               # the removed designs are gone from both builds
    assert ctx.lib.gpk_tune(ctx.h, 55, 4) < 0 and ctx.lib.gpk_tune(ctx.h, 55, 1) == 0 and ctx.lib.gpk_tune(ctx.h, 55, 0) == 0   # round 6: store policy 0 .. 3
    dev.close()
    mx, l2 = C.c_double(), C.c_double()
    one = ctx.array(np.array([3.0])); two = ctx.array(np.array([1.0]))
    assert ctx.lib.gpk_error_metrics(ctx.h, 1, one.ptr, two.ptr, None, C.byref(mx), C.byref(l2)) == 0
    assert mx.value == 2.0 and l2.value == 2.0
    assert ctx.lib.gpk_error_metrics(ctx.h, 0, one.ptr, two.ptr, None, C.byref(mx), C.byref(l2)) < 0
    assert ctx.lib.gpk_error_metrics(ctx.h, 1, None, two.ptr, None, C.byref(mx), C.byref(l2)) < 0
    ms = C.c_double()
    assert ctx.lib.gpk_prof_read_assembly(ctx.h, C.byref(ms)) < 0               # nothing timed yet
    # Darcy, leading-zero layout (default): N_data = 0 and N_boundary = 0 against the dense schedule
    rng = np.random.RandomState(8)
    for Nd, Nb, Ndata in ((150, 40, 0), (130, 0, 10)):
        Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
        Tu, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
        Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
        assert ctx.potrf(Tu) == 0 and ctx.potrf(Ta) == 0
        data = 0.1 * rng.normal(size=Ndata) if Ndata else np.zeros(0)
        z0 = 0.2 * rng.normal(size=6 * Nd)
        outs = []
        for mode in (1, 0):
            ctx.lib.gpk_debug_set(23, mode)
            try:
                prob = gpk.GNProblem(ctx, 'Darcy_flow2d', Nd, Nb, np.ones(Nd), np.zeros(Nb), Tu, p0=1e-2, data_u=data if Ndata else None, L2=Ta)
                z = ctx.array(z0)
                loss, info = ctx.gn_step(prob, z)
                assert info == 0 and np.isfinite(loss)
                outs.append((z.download().copy(), loss))
                prob.release_workspace()
            finally:
                ctx.lib.gpk_debug_set(23, 1)
        assert np.linalg.norm(outs[0][0] - outs[1][0]) <= 1e-9 * np.linalg.norm(outs[1][0])
        assert outs[0][1] == pytest.approx(outs[1][1], rel=1e-10)
    ctx.close()


def test_round5_entry_points_reject_bad_arguments_and_handle_edges():
    """gpk_mg_preflight without a communicator / with a buffer below one element / null outputs; gpk_tune keys of round 5 (52: the loss
    gpk_gn_step reports; 54 exists only in the development build); the exact in-step loss on a problem smaller than one 64-equation block and
    with the substitution schedule (no inverted blocks): same number as gpk_gn_loss."""
    import ctypes as C
    import gpk
    from oracle import gp_oracle as O
    ctx = gpk.Context(0)
    h = C.c_void_p()
    assert ctx.lib.gpk_mg_create(ctx.h, 0, 1, 512, C.byref(h)) == 0
    bms, ams, seen = (C.c_double * 1)(), C.c_double(), C.c_int()
    assert ctx.lib.gpk_mg_preflight(h, 1 << 20, 1, bms, C.byref(ams), C.byref(seen)) < 0          # no communicator bound
    assert 'no communicator' in ctx.lib.gpk_last_error(ctx.h).decode()
    assert ctx.lib.gpk_mg_preflight(h, 4, 1, bms, C.byref(ams), C.byref(seen)) < 0                # less than one double
    assert ctx.lib.gpk_mg_preflight(h, 1 << 20, 0, bms, C.byref(ams), C.byref(seen)) < 0          # no repetitions
    assert ctx.lib.gpk_mg_preflight(h, 1 << 20, 1, None, C.byref(ams), C.byref(seen)) < 0
    assert ctx.lib.gpk_mg_preflight(None, 1 << 20, 1, bms, C.byref(ams), C.byref(seen)) < 0
    assert ctx.lib.gpk_mg_destroy(h) == 0
    assert ctx.lib.gpk_tune(ctx.h, 54, 1) < 0 and 'development build' in ctx.lib.gpk_last_error(ctx.h).decode()
    assert ctx.lib.gpk_tune(ctx.h, 54, 0) == 0
    rng = np.random.RandomState(12)
    for Nd, Nb, dinv in ((20, 6, False), (20, 6, 256), (150, 0, False), (333, 41, 256)):
        Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2)) if Nb else np.zeros((0, 2))
        f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1]) if Nb else np.zeros(0)
        T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.3, Xd, Xb, 1e-8, 'adaptive')
        assert ctx.potrf(T) == 0
        prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, dinv=dinv)
        z0 = rng.normal(size=Nd)
        want = ctx.gn_loss(prob, ctx.array(z0))
        for mode in (1, 2):                                        # chain next to the end of the step / in front of the solve
            ctx.tune(52, mode)
            z = ctx.array(z0)
            got, info = ctx.gn_step(prob, z)
            assert info == 0 and got == pytest.approx(want, rel=1e-13), (Nd, Nb, dinv, mode)
        ctx.tune(52, 0)
        z = ctx.array(z0)
        approx, _ = ctx.gn_step(prob, z)                           # the free number of rounds 2-4: equal to a few digits beyond the tolerance of the loss histories
        assert approx == pytest.approx(want, rel=1e-6)
        ctx.tune(52, 1)
        prob.release_workspace(); T.free()
    ctx.close()

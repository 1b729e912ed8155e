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

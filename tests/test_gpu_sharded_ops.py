"""The multi-GPU schedule (gpk/sharded.py) on ONE GPU through the real block operations (GpuBlockOps: libgpk on torch
tensors' memory, explicit torch stream) -- world size 1 needs no process group.  Checks the panel Cholesky and the
Gauss-Newton step, in the plain and in the leading-zero (reversed) layout with its gpk_gn_build_rev / gpk_trsm_lz /
gpk_gemm_lz entry points, against the oracle.  (World sizes 2 and 3 run on CPU with a numpy test double:
tests/test_sharded_gloo.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O


def test_sharded_schedule_world1_on_gpu():
    import torch
    import gpk
    from gpk._lib import GNProblemStruct
    from gpk.sharded import Comm, GpuBlockOps, ShardedFactorSolve
    ctx = gpk.Context(0)
    ops = GpuBlockOps(ctx)
    solver = ShardedFactorSolve(ops, Comm(), nb=128)
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(11)
    Nd, Nb = 300, 60
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    sysm = O.EllipticSystem(1.0, 3.0, f, g)
    Theta = O.add_nugget(O.gram_matrix_assembly(Xd, Xb), 'Nonlinear_elliptic', Nd, Nb, 1e-7)[0]
    N, nz = 2 * Nd + Nb, Nd
    ld = ((N + 15) // 16) * 16
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    Lt = torch.zeros((N, ld), dtype=torch.float64, device=dev)
    Lt[:, :N] = t(Theta)
    assert solver.potrf(Lt, N) == 0
    Lref = O.cholesky(Theta)
    got = np.tril(Lt[:, :N].cpu().numpy())
    assert np.max(np.abs(got - Lref)) <= 1e-9 * np.max(np.abs(Lref))
    tf, tg = t(f), t(g)
    ps = GNProblemStruct()
    ps.system, ps.Nd, ps.Nb, ps.Ndata = 0, Nd, Nb, 0
    ps.p0, ps.p1, ps.pen_lambda = 1.0, 3.0, 0.0
    ps.rhs_f, ps.bdy_g, ps.data_u = tf.data_ptr(), tg.data_ptr(), None
    ps.L, ps.ldl, ps.L2, ps.ldl2 = Lt.data_ptr(), ld, None, 0
    lds = ((nz + 1 + 15) // 16) * 16
    S = torch.empty((N, lds), dtype=torch.float64, device=dev)
    Hb = torch.empty((nz + 1, lds), dtype=torch.float64, device=dev)
    delta = torch.empty(nz, dtype=torch.float64, device=dev)
    z0 = rng.normal(size=nz)
    sol_ref, hist_ref = O.gn_method(sysm, [Lref], z0, 3, 1)
    Dinv = ops.trtri_diag(Lt, N, block=256)                     # N > 256: several inverted blocks + a ragged last one
    for rev, dinv in ((False, False), (True, False), (False, True), (True, True)):
        z = t(z0)
        S2 = torch.zeros_like(S) if dinv else None
        hist = []
        for _ in range(3):
            loss_in, info = solver.gn_step(ps, nz, N, Lt, z, S, Hb, delta, 1.0, rev=rev, Dinv=Dinv if dinv else None, S2=S2)
            assert info == 0
            hist.append(loss_in)
        np.testing.assert_allclose(hist, hist_ref[:3], rtol=1e-6)
        zz = z.cpu().numpy()
        assert np.linalg.norm(zz - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref)
    torch.cuda.synchronize()
    ctx.close()

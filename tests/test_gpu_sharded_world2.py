"""The multi-rank schedule with the REAL block operations and more than one rank: two (and three) processes share the one
GPU of the test box, torch.distributed runs on gloo (host-staged collectives; NCCL refuses two ranks on one device).  What
this covers beyond the CPU gloo tests (numpy test double) and the world-size-1 GPU test: panels factored by one rank and
consumed by another as device memory, column shards of S solved by different processes with the leading-zero kernels and
stitched together, block rows of Hb from different ranks -- against the oracle."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    ROOT = {root!r}
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
    import torch, torch.distributed as dist
    import gpk
    from gpk._lib import GNProblemStruct
    from gpk.sharded import Comm, GpuBlockOps, ShardedFactorSolve
    from oracle import gp_oracle as O
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = gpk.Context(0)
    ops = GpuBlockOps(ctx)
    solver = ShardedFactorSolve(ops, Comm(), nb=128)
    solver.col_align = 64
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(21)
    Nd, Nb = 500, 80
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
    assert np.max(np.abs(got - Lref)) <= 1e-9 * np.max(np.abs(Lref)), 'panel-sharded Cholesky'
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
        assert np.linalg.norm(zz - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref), ('rev', rev)
        b = solver.column_ranges_lz(nz + 1, nz, N) if rev else None
        if rev:
            assert all(b[i] < b[i + 1] for i in range(world)), b          # every rank really owns a shard
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()
    print('rank', rank, 'ok')
''')


@pytest.mark.parametrize('world', [2, 3])
def test_real_block_ops_several_ranks_one_gpu(world, tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]
    assert all('ok' in o[0] for o in outs)

"""world_size-2 (and 3) CPU tests of the multi-GPU schedule (gpk/sharded.py) over the gloo backend.

The schedule -- block-cyclic panel ownership, panel broadcasts, column-sharded TRSM, all-gathers of S and Hb -- is the
code that runs under RCCL on the 8-GPU node; here its block operations are served by a numpy test double
(tests/_cpu_block_ops.py) so that correctness of the distributed algorithm is checked without GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gpk.sharded import Comm, ShardedFactorSolve
        from _cpu_block_ops import NumpyBlockOps
        from oracle import gp_oracle as O
        comm = Comm()
        rng = np.random.RandomState(7)                       # identical inputs on every rank
        if case == 'potrf':
            n, nb = 150, 32                                   # ragged last panel, 5 panels over `world` ranks
            M = rng.normal(size=(n, n))
            A0 = M @ M.T + n * np.eye(n)
            A = torch.from_numpy(np.ascontiguousarray(np.pad(A0, ((0, 0), (0, 10)))))        # ld > n
            solver = ShardedFactorSolve(NumpyBlockOps(), comm, nb=nb)
            assert solver.lookahead                               # default for more than one rank
            solver.trace = []
            info = solver.potrf(A, n)
            L = np.tril(A.numpy()[:, :n])
            np.save(os.path.join(out_dir, f'potrf_{rank}.npy'), L)
            assert info == 0
            # look-ahead order as ISSUED by this rank: the panel it owns next is updated and factored (and packed for its
            # broadcast) before the other columns it owns receive the current panel
            tr = solver.trace
            pos = {}
            for i, op in enumerate(tr):
                pos.setdefault(op[:3], i)
            nblk = (n + nb - 1) // nb
            for k in range(nblk - 1):
                if (k + 1) % world != rank:
                    continue
                assert pos[('UPDATE', k + 1, k)] < pos[('FACTOR', k + 1, 0)] < pos[('PACK', k + 1, 0)]
                later = [pos[('UPDATE', j, k)] for j in range(k + 2, nblk) if j % world == rank]
                assert all(pos[('PACK', k + 1, 0)] < p for p in later)
            assert [op[1] for op in tr if op[0] == 'BCAST'] == list(range(nblk))   # collectives in the same order on every rank
            # and the schedule without look-ahead gives the same factor bit for bit (same operations, same operands)
            A2 = torch.from_numpy(np.ascontiguousarray(np.pad(A0, ((0, 0), (0, 10)))))
            assert ShardedFactorSolve(NumpyBlockOps(), comm, nb=nb, lookahead=False).potrf(A2, n) == 0
            assert np.array_equal(np.tril(A2.numpy()[:, :n]), L)
            assert np.linalg.norm(L - np.linalg.cholesky(A0)) <= 1e-12 * np.linalg.norm(A0)
        elif case == 'potrf_bad':
            n, nb = 100, 32
            M = rng.normal(size=(n, n))
            A0 = M @ M.T + n * np.eye(n)
            A0[70, 70] = -5.0
            A = torch.from_numpy(A0.copy())
            info = ShardedFactorSolve(NumpyBlockOps(), comm, nb=nb).potrf(A, n)
            assert info == 64 + 1                             # every rank learns the failing panel's LAPACK index (its owner's device-side
                                                              # info word; the test double reports the first column of the failing block)
        elif case in ('gn', 'gn_shard_hb'):
            Nd, Nb, nb = 70, 20, 32
            Xd = rng.uniform(0, 1, (Nd, 2))
            Xb = rng.uniform(0, 1, (Nb, 2))
            f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1])
            g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
            sysm = O.EllipticSystem(1.0, 3.0, f, g)
            Theta = O.add_nugget(O.gram_matrix_assembly(Xd, Xb), 'Nonlinear_elliptic', Nd, Nb, 1e-6)[0]
            N, nz = 2 * Nd + Nb, Nd
 
            solver = ShardedFactorSolve(NumpyBlockOps(sysm), comm, nb=nb, shard_hb=(case == 'gn_shard_hb'))
            solver.col_align = 4                                  # small problem: let every rank own some columns
            Lt = torch.from_numpy(Theta.copy())
            assert solver.potrf(Lt, N) == 0
            z0 = rng.normal(size=nz)
            z = torch.from_numpy(z0.copy())
            S = torch.zeros((N, nz + 4), dtype=torch.float64)
            Hb = torch.zeros((nz + 1, nz + 4), dtype=torch.float64)
            delta = torch.zeros(nz, dtype=torch.float64)
            sol_ref, hist_ref = O.gn_method(sysm, [O.cholesky(Theta)], z0, 3, 1)
            Dinv = solver.ops.trtri_diag(Lt, N, block=32)         # (the test double takes any block size)
            # plain / leading-zero layout, substitution / inverted blocks; round 6: padded all-gathers / the DIRECT exchange (exact-size
            # point-to-point batches, what gpk_mg option key 3 = 2 does over ncclSend / ncclRecv) -- same numbers, the exchange moves bits
            finals = {}
            for rev, dinv, direct in ((False, False, False), (True, False, False), (False, True, False), (True, True, False),
                                      (True, True, True), (False, False, True)):
                solver.direct = direct
                z = torch.from_numpy(z0.copy())
                S2 = torch.zeros_like(S) if dinv else None
                hist = []
                for _ in range(3):
                    loss_in, info = solver.gn_step(None, nz, N, Lt, z, S, Hb, delta, 1.0, rev=rev, Dinv=Dinv if dinv else None, S2=S2)
                    assert info == 0
                    hist.append(loss_in)
                np.testing.assert_allclose(hist, hist_ref[:3], rtol=1e-7)
                assert np.linalg.norm(z.numpy() - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref)
                finals[(rev, dinv, direct)] = z.numpy().copy()
            assert np.array_equal(finals[(True, True, True)], finals[(True, True, False)])      # the exchange form does not change a bit
            assert np.array_equal(finals[(False, False, True)], finals[(False, False, False)])
            solver.direct = False
            np.save(os.path.join(out_dir, f'gn_{rank}.npy'), z.numpy())
            if case == 'gn':
                b = solver.column_ranges_lz(nz + 1, nz, N)        # work-balanced shards: monotone, cover everything
                assert b[0] == 0 and b[-1] == nz + 1 and all(b[i] < b[i + 1] for i in range(world))
                widths = np.diff(b)
                assert widths[0] > widths[-1]                     # early columns are cheap (long zero prefix): wider first shard
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, case, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)


@pytest.mark.parametrize('world', [2, 3])
def test_panel_sharded_cholesky(world, tmp_path):
    _run(world, 'potrf', tmp_path)
    Ls = [np.load(tmp_path / f'potrf_{r}.npy') for r in range(world)]
    for L in Ls[1:]:
        assert np.array_equal(L, Ls[0])                       # every rank ends with the same complete factor


def test_panel_sharded_cholesky_reports_failure(tmp_path):
    _run(2, 'potrf_bad', tmp_path)


def test_sharded_gauss_newton_step(tmp_path):
    _run(2, 'gn', tmp_path)
    z0, z1 = np.load(tmp_path / 'gn_0.npy'), np.load(tmp_path / 'gn_1.npy')
    assert np.array_equal(z0, z1)                             # replicated iterate, bit for bit


def test_sharded_gauss_newton_step_with_panel_sharded_hb(tmp_path):
    """the Cholesky of the bordered Gauss-Newton matrix through the panel plan as well (default from 4 ranks on)"""
    _run(3, 'gn_shard_hb', tmp_path)
    zs = [np.load(tmp_path / f'gn_{r}.npy') for r in range(3)]
    assert np.array_equal(zs[0], zs[1]) and np.array_equal(zs[0], zs[2])


def test_world_size_one_needs_no_process_group():
    from gpk.sharded import Comm, ShardedFactorSolve
    from _cpu_block_ops import NumpyBlockOps
    rng = np.random.RandomState(1)
    n = 90
    M = rng.normal(size=(n, n))
    A0 = M @ M.T + n * np.eye(n)
    A = torch.from_numpy(A0.copy())
    assert ShardedFactorSolve(NumpyBlockOps(), Comm(), nb=32).potrf(A, n) == 0
    assert np.linalg.norm(np.tril(A.numpy()) - np.linalg.cholesky(A0)) <= 1e-12 * np.linalg.norm(A0)

"""The native multi-GPU entry points (include/gpk_mg.h) through ctypes on the one GPU of the test box.

  * world 1: gpk_mg_gn_step IS gpk_gn_step (bit for bit); gpk_mg_potrf with the look-ahead plan forced on -- three HIP
    streams and the event dependencies of the plan on real hardware -- against the oracle's factor, LAPACK info on a bad pivot;
    the RCCL binding (dlopen of the library torch uses, ncclGetUniqueId, ncclCommInitRank with one rank) comes up;
  * world 2 and 3: several processes share the GPU (RCCL refuses that), so ncclBroadcast / ncclAllGather are replaced by
    host-staged stand-ins over gloo bound to the SAME entry points (gpk.mg.MultiGpu(comm='staged')): the native schedule --
    plan executor with look-ahead, column shards, all-gathers, replicated and panel-sharded Cholesky of Hb -- runs unchanged and
    must reproduce the oracle's Gauss-Newton iterates, identically on every rank.
"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(Nd, Nb, seed, nugget=1e-7):
    rng = np.random.RandomState(seed)
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    Theta = O.add_nugget(O.gram_matrix_assembly(Xd, Xb), 'Nonlinear_elliptic', Nd, Nb, nugget)[0]
    return Xd, Xb, f, g, Theta, rng.normal(size=Nd)


def test_world1_gn_step_is_gpk_gn_step_bit_for_bit():
    import gpk
    from gpk.mg import MultiGpu
    Nd, Nb = 700, 100
    Xd, Xb, f, g, Theta, z0 = _problem(Nd, Nb, 5)
    ctx = gpk.Context(0)
    T = ctx.array(Theta)
    assert ctx.potrf(T) == 0
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
    S, H, delta, _ = prob.workspace()
    za, zb = ctx.array(z0), ctx.array(z0)
    mgpu = MultiGpu(ctx, 0, 1, panel=256)
    S2 = ctx.empty(S.rows, S.cols, S.ld); S2.zero()
    for _ in range(3):
        la, ia = ctx.gn_step(prob, za)
        lb, ib = mgpu.gn_step(prob.struct, zb.ptr, 1.0, S.ptr, S.ld, S2.ptr, H.ptr, H.ld, delta.ptr)
        assert la == lb and ia == ib == 0
        assert np.array_equal(za.download(), zb.download())
    mgpu.close()
    ctx.close()


@pytest.mark.parametrize('lookahead', [0, 1])
def test_world1_potrf_plan_executor(lookahead):
    import gpk
    from gpk.mg import MultiGpu
    Nd, Nb = 900, 124                                             # order 1924: 8 panels of 256 (ragged last one)
    _, _, _, _, Theta, _ = _problem(Nd, Nb, 6)
    N = Theta.shape[0]
    ctx = gpk.Context(0)
    mgpu = MultiGpu(ctx, 0, 1, panel=256)
    mgpu.set_option('lookahead', lookahead)
    Lref = O.cholesky(Theta)
    for rep in range(3):                                          # repeated: events and streams are reused
        A = ctx.array(Theta)
        assert mgpu.potrf(A.ptr, N, A.ld) == 0
        got = np.tril(A.download())
        assert np.max(np.abs(got - Lref)) <= 1e-9 * np.max(np.abs(Lref))
        A.free()
    bad = Theta.copy()
    bad[1500, 1500] = -1.0                                        # a non-positive pivot in the sixth panel
    A = ctx.array(bad)
    assert mgpu.potrf(A.ptr, N, A.ld) == 1501
    mgpu.close()
    ctx.close()


def test_rccl_binding_comes_up_with_one_rank():
    """dlopen of the RCCL library this process uses, ncclGetUniqueId, ncclCommInitRank(nranks = 1) on the handle's device, the two
    collectives bound: what bench.py does on every rank of the 8-GPU run, minus the peers."""
    import ctypes as C
    import gpk
    from gpk.mg import loaded_hip_runtime, torch_rccl_path
    ctx = gpk.Context(0)
    path = torch_rccl_path()                                      # the librccl of the ROCm stack this process runs on (see its docstring)
    assert path is not None and os.path.dirname(path) == os.path.dirname(loaded_hip_runtime())
    print('\n[rccl] HIP runtime', loaded_hip_runtime(), '-> RCCL', path)
    uid = (C.c_char * 128)()
    assert ctx.lib.gpk_mg_rccl_unique_id(path.encode(), uid) == 0
    assert any(b != 0 for b in uid.raw)
    h = C.c_void_p()
    assert ctx.lib.gpk_mg_create(ctx.h, 0, 1, 512, C.byref(h)) == 0
    rc = ctx.lib.gpk_mg_rccl_init(h, path.encode(), uid)
    assert rc == 0, ctx.lib.gpk_last_error(ctx.h).decode()
    ok = C.c_int()
    rc = ctx.lib.gpk_mg_selftest(h, C.byref(ok))                  # the REAL ncclBroadcast / ncclAllGather, one rank: argument order, type codes
    assert rc == 0 and ok.value == 1, ctx.lib.gpk_last_error(ctx.h).decode()
    bms, ams, seen = (C.c_double * 1)(), C.c_double(), C.c_int()  # (round 5) gpk_mg_preflight through the real RCCL entry points, one rank
    rc = ctx.lib.gpk_mg_preflight(h, 64 << 20, 2, bms, C.byref(ams), C.byref(seen))
    assert rc == 0 and seen.value == 1 and bms[0] > 0 and ams.value > 0, ctx.lib.gpk_last_error(ctx.h).decode()
    print(f'[rccl] preflight, one rank: broadcast of 64 MB {bms[0]:.3f} ms, all-gather {ams.value:.3f} ms')
    # (round 6) the point-to-point entry points of the REAL library are bound by gpk_mg_rccl_init (ncclSend / ncclRecv / ncclGroupStart /
    # ncclGroupEnd by dlsym) and an (empty, one rank) group goes through them; the escape hatch switches them off
    assert ctx.lib.gpk_mg_has_p2p(h) == 1
    pms = C.c_double()
    rc = ctx.lib.gpk_mg_preflight_p2p(h, 64 << 20, 2, C.byref(pms))
    assert rc == 0 and pms.value >= 0.0, ctx.lib.gpk_last_error(ctx.h).decode()
    assert ctx.lib.gpk_mg_set_option(h, 3, 2) == 0                  # direct exchange selectable ...
    assert ctx.lib.gpk_mg_set_option(h, 4, 0) == 0 and ctx.lib.gpk_mg_has_p2p(h) == 0
    assert ctx.lib.gpk_mg_set_option(h, 3, 2) < 0                   # ... and refused once the entry points are switched off
    assert ctx.lib.gpk_mg_destroy(h) == 0
    ctx.close()


def test_rccl_communicator_next_to_torchs_nccl_process_group(tmp_path):
    """What every rank of `bench.py --gpus N` does before the sharded run, with N = 1: torch.distributed on the nccl backend (its own
    RCCL communicator), the unique id through broadcast_object_list on that group, a SECOND communicator created by libgpk from the
    same RCCL library, the self-test through ncclBroadcast / ncclAllGather, then a factorisation and a step through gpk_mg_*."""
    script = tmp_path / 'one_rank.py'
    script.write_text(textwrap.dedent(f'''
        import os, sys
        import numpy as np
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'nonlinpdes-gpsolver_amd'))
        import torch, torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group('nccl')
        t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()      # torch's communicator exists
        import gpk
        from gpk.mg import MultiGpu
        from oracle import gp_oracle as O
        ctx = gpk.Context(0)
        mgpu = MultiGpu(ctx, 0, 1, panel=256, comm='rccl')
        assert 'torch/lib/librccl' in mgpu.comm_kind, mgpu.comm_kind
        assert mgpu.selftest()
        rng = np.random.RandomState(3)
        Nd, Nb = 400, 80
        Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
        Theta = O.add_nugget(O.gram_matrix_assembly(Xd, Xb), 'Nonlinear_elliptic', Nd, Nb, 1e-7)[0]
        T = ctx.array(Theta)
        assert mgpu.potrf(T.ptr, 2 * Nd + Nb, T.ld) == 0
        assert np.max(np.abs(np.tril(T.download()) - O.cholesky(Theta))) <= 1e-9 * np.max(np.abs(Theta))
        dist.barrier(); dist.destroy_process_group()
        mgpu.close(); ctx.close()
        print('ok')
    '''))
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'ok' in p.stdout, p.stderr[-3000:]


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    ROOT = {root!r}
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
    import torch, torch.distributed as dist
    import gpk
    from gpk.mg import MultiGpu
    from oracle import gp_oracle as O
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = gpk.Context(0)
    rng = np.random.RandomState(21)
    Nd, Nb = 500, 80
    Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    sysm = O.EllipticSystem(1.0, 3.0, f, g)
    Theta = O.add_nugget(O.gram_matrix_assembly(Xd, Xb), 'Nonlinear_elliptic', Nd, Nb, 1e-7)[0]
    N, nz = 2 * Nd + Nb, Nd
    Lref = O.cholesky(Theta)
    z0 = rng.normal(size=nz)
    sol_ref, hist_ref = O.gn_method(sysm, [Lref], z0, 3, 1)
    mgpu = MultiGpu(ctx, rank, world, panel=128, comm='staged')
    assert mgpu.selftest()
    mgpu.set_option('p2p', 0); assert not mgpu.has_p2p()          # (round 6) the escape hatch of the direct exchange (GPK_MG_P2P=0)
    mgpu.set_option('p2p', 1); assert mgpu.has_p2p()
    pf = mgpu.preflight(1 << 20, 1)                               # (round 5) the bandwidth preflight bench.py runs before a sharded run
    assert pf['ranks_seen_by_rccl'] == world and len(pf['bcast_ms_by_root']) == world
    assert all(m > 0 for m in pf['bcast_ms_by_root']) and pf['allgather_ms'] > 0 and all(v > 0 for v in pf['bcast_gbs_by_root'])
    mgpu.set_option('col_align', 64)
    results = []
    assert pf['p2p_bound'] and pf['direct_ms'] > 0                 # (round 6) the direct exchange: grouped send / recv stand-ins
    for lookahead, shard_hb, overlap_s in ((0, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1), (1, 0, 2), (1, 1, 2), (1, 0, -1)):
        mgpu.set_option('lookahead', lookahead)
        mgpu.set_option('shard_hb', shard_hb)
        mgpu.set_option('overlap_s', overlap_s)                   # shards of S: one all-gather / one broadcast per shard, products chasing
        T = ctx.array(Theta)
        assert mgpu.potrf(T.ptr, N, T.ld) == 0
        got = np.tril(T.download())
        assert np.max(np.abs(got - Lref)) <= 1e-9 * np.max(np.abs(Lref)), ('panel-sharded Cholesky', lookahead)
        prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0, dinv=256)
        S, H, delta, _ = prob.workspace()
        S2 = ctx.empty(S.rows, S.cols, S.ld); S2.zero()
        z = ctx.array(z0)
        hist = []
        for _ in range(3):
            loss, info = mgpu.gn_step(prob.struct, z.ptr, 1.0, S.ptr, S.ld, S2.ptr, H.ptr, H.ld, delta.ptr)
            assert info == 0
            hist.append(loss)
        np.testing.assert_allclose(hist, hist_ref[:3], rtol=1e-6)
        zz = z.download()
        assert np.linalg.norm(zz - sol_ref) <= 1e-7 * np.linalg.norm(sol_ref), (lookahead, shard_hb, overlap_s)
        results.append(zz)
        prob.release_workspace()
    # a non-positive pivot in a panel of rank 1: every rank reports the same LAPACK index
    bad = Theta.copy(); bad[200, 200] = -1.0
    T = ctx.array(bad)
    assert mgpu.potrf(T.ptr, N, T.ld) == 201
    np.save(os.path.join({out!r}, f'z_{{rank}}.npy'), np.stack(results))
    dist.barrier()
    mgpu.close()
    dist.destroy_process_group()
    ctx.close()
    print('rank', rank, 'ok')
''')


@pytest.mark.parametrize('world', [2, 3])
def test_native_schedule_several_ranks_one_gpu(world, tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]
    assert all('ok' in o[0] for o in outs)
    zs = [np.load(tmp_path / f'z_{r}.npy') for r in range(world)]
    for z in zs[1:]:
        assert np.array_equal(z, zs[0])                               # replicated iterate, bit for bit, in every mode


BIG_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    ROOT = {root!r}
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
    import torch, torch.distributed as dist
    import gpk
    from gpk.mg import MultiGpu
    from oracle import gp_oracle as O
    from src.sample_points import sampled_pts_rdm
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = gpk.Context(0)
    np.random.seed(0)
    Nd, Nb = 4000, 400                                            # BASELINE config 2: Theta of order 8400, 17 panels of 512
    Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
    z0 = np.random.normal(0.0, 1.0, Nd)
    f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
    N = 2 * Nd + Nb
    T1, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-12, 'adaptive')
    T2 = T1.clone()
    assert ctx.potrf(T1) == 0                                     # the single-GPU factorisation
    want = np.tril(T1.download())
    mgpu = MultiGpu(ctx, rank, world, panel=512, comm='staged')
    mgpu.set_option('lookahead', 1)                               # the plan the 8-GPU run uses: three streams, events, broadcasts
    assert mgpu.potrf(T2.ptr, N, T2.ld) == 0
    got = np.tril(T2.download())
    same = np.array_equal(got, want)
    dev = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    print('rank', rank, 'look-ahead plan vs gpk_potrf: bitwise', same, 'max rel dev', dev, flush=True)
    assert same, dev
    # the step: native sharded (both exchange forms) against the single-GPU step on the same factor
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T2, p0=1.0, p1=3.0)
    S, H, delta, _ = prob.workspace()
    S2 = ctx.empty(S.rows, S.cols, S.ld); S2.zero()
    zs = []
    for overlap in (0, 1):
        mgpu.set_option('overlap_s', overlap)
        z = ctx.array(z0)
        for _ in range(2):
            loss, info = mgpu.gn_step(prob.struct, z.ptr, 1.0, S.ptr, S.ld, S2.ptr, H.ptr, H.ld, delta.ptr)
            assert info == 0
        zs.append(z.download().copy())
    assert np.array_equal(zs[0], zs[1])                           # the exchange form does not change a bit
    z = ctx.array(z0)
    for _ in range(2):
        ctx.gn_step(prob, z)
    one = z.download()
    rel = float(np.linalg.norm(zs[0] - one) / np.linalg.norm(one))
    print('rank', rank, 'sharded step vs one-GPU step: rel dev', rel, flush=True)
    assert rel <= 1e-9
    np.save(os.path.join({out!r}, f'zbig_{{rank}}.npy'), zs[0])
    dist.barrier()
    mgpu.close()
    dist.destroy_process_group()
    ctx.close()
    print('rank', rank, 'ok')
''')


def test_native_lookahead_plan_world2_at_config2_size_is_gpk_potrf_bit_for_bit(tmp_path):
    """Round 4 (verdict item 4c): the NATIVE executor with the look-ahead plan, two ranks (staged collectives on the one GPU of the
    box), at BASELINE config 2's size -- Theta of order 8400, 17 panels of 512, every update through the kernels the real run uses --
    must reproduce gpk_potrf bit for bit on both ranks; the sharded step with either exchange form must give the same bits and agree
    with the one-GPU step."""
    script = tmp_path / 'big_worker.py'
    script.write_text(BIG_WORKER.format(root=ROOT, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(o[0][-1500:], o[1][-3000:]) for o in outs]
    assert all('ok' in o[0] for o in outs)
    assert np.array_equal(np.load(tmp_path / 'zbig_0.npy'), np.load(tmp_path / 'zbig_1.npy'))


SYS_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    ROOT = {root!r}
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
    import torch, torch.distributed as dist
    import gpk
    from gpk.mg import MultiGpu
    from oracle import gp_oracle as O
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = gpk.Context(0)
    mgpu = MultiGpu(ctx, rank, world, panel=128, comm='staged')
    mgpu.set_option('col_align', 64)
    out = {{}}
    for system in ('Burgers', 'Eikonal', 'Darcy_flow2d'):
        rng = np.random.RandomState(31)
        if system == 'Darcy_flow2d':
            Nd, Nb, Ndata = 230, 44, 17
            Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
            f = np.ones(Nd); g = np.zeros(Nb)
            data = 0.05 * np.sin(np.pi * Xd[:Ndata, 0]) * np.sin(np.pi * Xd[:Ndata, 1]) + 1e-3 * rng.normal(size=Ndata)
            T, _ = ctx.assemble('Darcy_u', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
            Ta, _ = ctx.assemble('Darcy_a', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
            assert ctx.potrf(Ta) == 0
            kw = dict(p0=1e-3, data_u=data, L2=Ta); sysm = O.DarcySystem(f, g, data, 1e-3); steps = 3
            z0 = 0.3 * rng.normal(size=6 * Nd)
        elif system == 'Burgers':
            Nd, Nb = 330, 61
            Xd = np.stack([rng.uniform(0, 1, Nd), rng.uniform(-1, 1, Nd)], axis=1)
            Xb = np.stack([rng.uniform(0, 1, Nb), rng.uniform(-1, 1, Nb)], axis=1)
            f = np.zeros(Nd); g = -np.sin(np.pi * Xb[:, 1]) * (rng.uniform(size=Nb) < 0.4)
            T, _ = ctx.assemble('Burgers', 'anisotropic_Gaussian', [0.3, 0.05], Xd, Xb, 1e-5, 'adaptive')
            kw = dict(p0=1.0, p1=0.02); sysm = O.BurgersSystem(1.0, 0.02, f, g); steps = 2
            z0 = 0.2 * rng.normal(size=3 * Nd)
        else:
            Nd, Nb = 350, 70
            Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
            f = np.ones(Nd); g = np.zeros(Nb)
            T, _ = ctx.assemble('Eikonal', 'Gaussian', 0.2, Xd, Xb, 1e-6, 'adaptive')
            kw = dict(p0=0.1); sysm = O.EikonalSystem(0.1, f, g); steps = 3
            z0 = 0.1 * rng.normal(size=3 * Nd)
        assert ctx.potrf(T) == 0
        L = np.tril(T.download())
        Ls = [np.tril(Ta.download()), L] if system == 'Darcy_flow2d' else [L]       # (oracle order: [L_a, L_u])
        sol_ref, hist_ref = O.gn_method(sysm, Ls, z0, steps, 1)
        prob = gpk.GNProblem(ctx, system, Nd, Nb, f, g, T, dinv=256, **kw)
        one = ctx.array(z0)                                       # the one-GPU step of the same problem
        for _ in range(steps):
            ctx.gn_step(prob, one)
        one = one.download()
        S, H, delta, _ = prob.workspace()
        S2 = ctx.empty(S.rows, S.cols, S.ld)
        res = []
        for shard_hb, overlap_s in ((0, 0), (0, 1), (1, 2), (0, -1)):
            mgpu.set_option('shard_hb', shard_hb); mgpu.set_option('overlap_s', overlap_s)
            S2.zero()
            z = ctx.array(z0)
            hist = []
            for _ in range(steps):
                loss, info = mgpu.gn_step(prob.struct, z.ptr, 1.0, S.ptr, S.ld, S2.ptr, H.ptr, H.ld, delta.ptr)
                assert info == 0
                hist.append(loss)
            zz = z.download()
            np.testing.assert_allclose(hist, hist_ref[:steps], rtol=1e-6)
            assert np.linalg.norm(zz - sol_ref) <= 1e-6 * np.linalg.norm(sol_ref), (system, shard_hb, overlap_s)
            assert np.linalg.norm(zz - one) <= 1e-8 * np.linalg.norm(one), (system, shard_hb, overlap_s)
            res.append(zz)
        out[system] = np.stack(res)
        prob.release_workspace()
    np.savez(os.path.join({out!r}, f'sys_{{rank}}.npz'), **out)
    dist.barrier()
    mgpu.close()
    dist.destroy_process_group()
    ctx.close()
    print('rank', rank, 'ok')
''')


@pytest.mark.parametrize('world', [2, 3])
def test_native_sharded_step_of_the_burgers_eikonal_and_darcy_systems(world, tmp_path):
    """Round 6: gpk_mg_gn_step for the Burgers (staircase of slope 1/3), Eikonal (two-segment profile) and Darcy (u-part sharded under its
    three-segment profile, cached a-part, data rows) systems -- column shards cut by work under their own profiles, every exchange form,
    replicated and panel-sharded Cholesky of Hb -- against the oracle (<= 1e-6), the one-GPU step (<= 1e-8) and across ranks (bit for bit).
    Several ranks on ONE GPU, host-staged collectives."""
    script = tmp_path / 'worker.py'
    script.write_text(SYS_WORKER.format(root=ROOT, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]
    zs = [np.load(tmp_path / f'sys_{r}.npz') for r in range(world)]
    for z in zs[1:]:
        for k in ('Burgers', 'Eikonal', 'Darcy_flow2d'):
            assert np.array_equal(z[k], zs[0][k])

"""Control flow of bench.py with N > 1 ranks, on CPU: two gloo processes, the GPU-touching functions replaced by stubs.
Checks what the driver relies on: every rank takes part in both phases (replicas of config 2, then the sharded config 5),
exactly ONE JSON line comes out (rank 0), the compact one (bench_line.py; the full object goes to bench_detail.json); since round 4
its `value` is the SHARDED config 5 (strong scaling, `value_workload` = "c5") with the same configuration timed on rank 0 alone beside
it (vs_1gpu) and the config-2 replicas as a secondary object -- and when the sharded run fails on one side, a line survives with the
error attached, `value` NULL (never another workload's figure under the same key, round 5) and the replicas under replicas_c2."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_flow_prints_one_line(tmp_path):
    script = tmp_path / 'drive.py'
    script.write_text(textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'nonlinpdes-gpsolver_amd'))
        import bench
        def fake_single(args, workload, comm=None):
            assert workload == 'c2' and comm is not None and comm.world == 2
            comm.barrier()
            t = comm.max_float(0.5 + comm.rank, 'cpu')           # slowest rank counts
            out = {{'metric': 'm', 'value': comm.world * args.steps / t, 'n_gpus': comm.world, 'scaling': 'weak', 'steps': args.steps}}
            return out if comm.rank == 0 else None
        def fake_sharded(args, workload, steps=None, warmup=None, solo=False):
            assert workload == 'c5'
            import torch.distributed as dist
            if solo:                                             # rank 0 alone: the 1-GPU point of the same job
                assert dist.get_rank() == 0 and steps == min(args.steps, 3) and warmup == 1
                return dict({{k: 1 for k in ('unit', 'steps', 'warmup', 'ms_per_step', 'config', 'l2_error', 'f1_tflops', 'one_time_ms', 'roofline')}},
                            value=2.0, n_gpus=1, scaling='strong')
            assert steps is None and warmup is None
            dist.barrier()
            if dist.get_rank() != 0:
                return None
            return dict({{k: 1 for k in ('unit', 'steps', 'warmup', 'ms_per_step', 'config', 'l2_error', 'f1_tflops', 'one_time_ms', 'roofline')}},
                        metric='m', value=3.0, n_gpus=2, scaling='strong')
        bench.run_single, bench.run_sharded = fake_single, fake_sharded
        sys.argv = ['bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1']
        bench.main()
    '''))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), GPK_BENCH_BACKEND='gloo', GPK_BENCH_DETAIL_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = [l for o in outs for l in o[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['value'] == 3.0      # the sharded configuration IS the value
    assert d['value_workload'] == 'c5' and len(lines[0]) < 7000
    full = json.load(open(tmp_path / 'bench_detail.json'))       # the full object beside it
    assert full['value'] == 3.0 and 'scaling_series' in full
    assert d['vs_1gpu'] == 1.5 and d['parallel_efficiency'] == 0.75 and d['one_gpu_same_job']['n_gpus'] == 1
    assert d['replicas_c2']['scaling'] == 'weak'
    assert abs(d['replicas_c2']['value'] - 2 * 4 / 1.5) < 1e-4   # max over ranks of (0.5, 1.5); the line carries 6 significant digits
    assert abs(full['replicas_c2']['value'] - 2 * 4 / 1.5) < 1e-9
    assert d['parity_failed'] is None


def test_two_rank_flow_survives_a_one_sided_failure(tmp_path):
    """Rank 1 fails in the middle of the sharded run while rank 0 is blocked in a collective: rank 0 must still print the
    primary value (with the error attached) and both ranks must leave promptly with exit code 0."""
    script = tmp_path / 'drive.py'
    script.write_text(textwrap.dedent(f'''
        import os, sys, time
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'nonlinpdes-gpsolver_amd'))
        import bench
        def fake_single(args, workload, comm=None):
            comm.barrier()
            out = {{'metric': 'm', 'value': 42.0, 'n_gpus': comm.world, 'scaling': 'weak', 'steps': args.steps}}
            return out if comm.rank == 0 else None
        def fake_sharded(args, workload, steps=None, warmup=None, solo=False):
            import torch.distributed as dist
            if dist.get_rank() == 1:
                time.sleep(1.0)
                raise RuntimeError('libgpk error -2: out of memory (simulated)')
            dist.barrier()                                       # rank 0 waits for a peer that never arrives
            return None
        bench.run_single, bench.run_sharded = fake_single, fake_sharded
        sys.argv = ['bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1']
        bench.main()
    '''))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), GPK_BENCH_BACKEND='gloo', GPK_BENCH_DETAIL_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = [l for o in outs for l in o[0].splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['value'] is None and d['value_workload'] == 'c5' and d['n_gpus'] == 2 and 'did not complete' in d['fallback']
    assert d['replicas_c2']['value'] == 42.0 and d['replicas_c2']['scaling'] == 'weak'
    assert 'out of memory (simulated)' in d['sharded_config']['error'] and 'rank 1' in d['sharded_config']['error']


def test_flop_models_of_the_roofline_objects():
    """bench.py divides MODELLED flop counts by measured times: the models must sit between the exact structural count and the
    dense count, and reproduce the launch count seen in the kernel trace (17 NN GEMM launches per step at config 2)."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module('bench')
    N, nz = 8400, 4000
    flops, launches = bench.trsm_dinv_executed_flops(N, nz, 1024)
    dense = float(N) * N * (nz + 1)
    exact = (N ** 3 - (N - nz) ** 3) / 3.0                        # sum over columns of (rows below the first non-zero)^2
    assert launches == 17
    assert exact <= flops <= 1.03 * exact and flops < 0.62 * dense
    syrk = bench.syrk_executed_flops(N, nz)
    assert bench.syrk_pipelined_flops(N, nz) == syrk              # the pipelined launches skip the tiles above the diagonal
    assert 0.69 * N * (nz + 1) ** 2 < syrk < 0.71 * N * (nz + 1) ** 2
    f256, l256 = bench.trsm_dinv_executed_flops(N, nz, 256)
    assert l256 > launches and abs(f256 - flops) < 0.02 * flops   # same work, more (smaller) launches


def test_hanging_sharded_run_is_abandoned_at_the_deadline(tmp_path):
    """One process, no process group: the secondary run hangs; at the deadline the primary line is printed with the reason and
    the process exits 0."""
    script = tmp_path / 'drive.py'
    script.write_text(textwrap.dedent(f'''
        import os, sys, time
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'nonlinpdes-gpsolver_amd'))
        import bench
        bench.run_single = lambda args, workload, comm=None, **kw: {{'metric': 'm', 'value': 7.0, 'n_gpus': 1, 'steps': args.steps}}
        def hang(args, workload, steps=None, warmup=None, solo=False):
            time.sleep(3600)
        bench.run_sharded = hang
        sys.argv = ['bench.py', '--steps', '4', '--warmup', '1', '--no-n10k', '--no-c3c4']
        bench.main()
    '''))
    env = dict(os.environ, GPK_SHARDED_TIMEOUT='2', GPK_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'GPK_FORCE_PG'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['value'] == 7.0 and 'deadline' in d['sharded_config']['error']


# ------------------------------------------------------------------------------------------------------------------ round 6
# The BARE command `python3 bench.py --gpus N ...` (no torchrun around it): bench.py starts its own N ranks (bench_launch.py), the
# parent never touches the GPU, rank 0's compact line is the parent's last stdout line, exit code = worst rank.
STUBS = '''
def run_single(args, workload, comm=None, secondary=False, steps=None, warmup=None):
    assert workload == 'c2' and comm is not None and comm.world == int(os.environ['WORLD_SIZE'])
    comm.barrier()
    out = {'metric': 'm', 'value': 10.0 * comm.world, 'unit': 'GN steps/s', 'n_gpus': comm.world, 'scaling': 'weak', 'steps': args.steps}
    return out if comm.rank == 0 else None
def run_sharded(args, workload, steps=None, warmup=None, solo=False):
    import torch.distributed as dist
    assert workload == 'c5' and os.environ.get('GPK_BENCH_SELF_LAUNCHED') == '1'
    base = {k: 1 for k in ('unit', 'warmup', 'ms_per_step', 'l2_error', 'f1_tflops', 'one_time_ms', 'roofline')}
    if solo:
        return dict(base, value=2.0, n_gpus=1, scaling='strong', steps=steps, config={'workload': 'c5 stub'}, ms_per_step=300.0,
                    one_time_ms={'cholesky_theta_sharded': 250.0}, cholesky_hb_alone_ms=36.0)
    dist.barrier()
    if dist.get_rank() != 0:
        return None
    return dict(base, metric='m', value=3.0, n_gpus=dist.get_world_size(), scaling='strong', steps=args.steps, config={'workload': 'c5 stub'},
                preflight={'ranks_seen_by_rccl': dist.get_world_size(), 'bcast_gbs_by_root': [1.0] * dist.get_world_size()})
'''


def _bare(args, env, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bare_command_launches_its_own_ranks(tmp_path):
    stubs = tmp_path / 'stubs.py'
    stubs.write_text(STUBS)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GPK_BENCH_BACKEND='gloo', GPK_BENCH_COMM='staged', GPK_BENCH_STUBS=str(stubs), GPK_BENCH_DETAIL_DIR=str(tmp_path))
    r = _bare(['--gpus', '2', '--steps', '1', '--warmup', '0'], env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]      # ONE line, and it is the last thing on stdout
    assert len(lines[0]) < 8192
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value_workload'] == 'c5' and d['value'] == 3.0 and d['scaling'] == 'strong'
    assert d['one_gpu_same_job']['value'] == 2.0 and d['vs_1gpu'] == 1.5
    assert d['preflight']['ranks_seen_by_rccl'] == 2
    assert d['replicas_c2']['value'] == 20.0
    assert d['data'].startswith('STUBBED')                        # a stubbed line can never be mistaken for a measurement
    # the host model beside the measurement: per rank count, expected ms of every multi-GPU choice, and the predicted ratio
    for P in ('2', '4', '8'):
        assert len(d['predicted'][P]['chol']) == 2 and len(d['predicted'][P]['xchg']) == 3 and d['predicted'][P]['x'] > 1
    assert d['predicted_vs_1gpu'] == pytest.approx(d['predicted']['2']['x'], abs=0.01)


def test_bare_command_fails_loudly_without_gpus(tmp_path):
    """no stubs, no GPU in this container: over RCCL nothing is launched (device count < N, exit 2); over gloo the ranks start, the
    product refuses to run without the HIP runtime, every rank exits non-zero and the parent passes that on -- no JSON line, no hang"""
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip('needs a box without GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'GPK_BENCH_STUBS')}
    env.update(GPK_BENCH_DETAIL_DIR=str(tmp_path))
    r = _bare(['--gpus', '2', '--steps', '1', '--warmup', '0'], dict(env, GPK_BENCH_BACKEND='nccl'))
    assert r.returncode == 2 and 'needs 2 visible GPUs' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]
    r = _bare(['--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'], dict(env, GPK_BENCH_BACKEND='gloo', GPK_BENCH_PEER_GRACE='20'))
    assert r.returncode != 0 and 'rank exit codes' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_launcher_relays_worst_exit_code_and_ends_stuck_peers(tmp_path):
    sys.path.insert(0, ROOT)
    import io
    import bench_launch
    script = tmp_path / 'child.py'
    script.write_text(textwrap.dedent('''
        import os, sys, time
        r = int(os.environ['RANK'])
        assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
        if r == 0:
            print('library banner'); print('{"value": 1}'); print('{"value": 2}', flush=True); time.sleep(600)   # stuck peer
        if r == 1:
            print('noise from rank 1', flush=True); sys.exit(7)
        time.sleep(600)
    '''))
    out, err = io.StringIO(), io.StringIO()
    t0 = __import__('time').monotonic()
    rc = bench_launch.launch(3, argv=[], script=str(script), env=dict(os.environ, GPK_BENCH_BACKEND='gloo', GPK_BENCH_PEER_GRACE='2'),
                             out=out, err=err)
    assert __import__('time').monotonic() - t0 < 60
    assert rc >= 7                                                # rank 1's code, or a signal code of the ranks that had to be ended
    lines = out.getvalue().splitlines()
    assert lines[-1] == '{"value": 2}' and lines[0] == 'library banner'
    assert '[rank 1 stdout] noise from rank 1' in err.getvalue() and 'did not leave' in err.getvalue()
    assert bench_launch.local_rank_of(3, 4, 1, 'gloo') == 0 and bench_launch.local_rank_of(3, 4, 8, 'nccl') == 3


def test_multi_gpu_model_is_consistent():
    """bench_model.py: pure host.  One rank reproduces the measured 1-GPU times it was fitted to; the shard rule is the library's; more
    ranks never predict a slower factorisation with look-ahead than without; the padded all-gather loses to exact broadcasts exactly
    when (P - 1) x widest shard exceeds the whole of S; a faster fabric predicts a faster step."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
    import bench_model as M
    from gpk.mg import column_bounds
    for P in (1, 2, 3, 4, 8):
        assert M.column_bounds(16001, 16000, 34000, P) == [int(v) for v in column_bounds(16001, 16000, 34000, P)]
        assert M.column_bounds(901, 900, 1924, P, 64) == [int(v) for v in column_bounds(901, 900, 1924, P, 64)]
    one = M.predict(34000, 16000, 1)
    assert one['cholesky_theta_ms']['sequential'] == pytest.approx(M.DEFAULT_ONE_GPU['cholesky_theta_ms'], rel=1e-9)
    assert one['step_ms_best'] == pytest.approx(M.DEFAULT_ONE_GPU['step_ms'], rel=1e-9) and one['predicted_vs_1gpu'] == pytest.approx(1.0)
    t = M.table()
    prev = None
    for P in (2, 4, 8):
        p = M.predict(34000, 16000, P)
        assert p['cholesky_theta_ms']['lookahead'] <= p['cholesky_theta_ms']['sequential']
        widest, total = max(p['shard_widths']), sum(p['shard_widths'])
        assert (p['exchange_of_S_ms']['all_gather_padded'] > p['exchange_of_S_ms']['broadcasts_exact']) == ((P - 1) * widest > total) or P == 2
        assert 1.0 < p['predicted_vs_1gpu'] < P and (prev is None or p['step_ms_best'] < prev)
        prev = p['step_ms_best']
        assert t[str(P)]['predicted_vs_1gpu'] == p['predicted_vs_1gpu']
        fast = M.predict(34000, 16000, P, fabric={'bcast_gbs': 600.0, 'allgather_gbs': 600.0})
        assert fast['step_ms_best'] < p['step_ms_best'] and fast['cholesky_theta_ms']['lookahead'] <= p['cholesky_theta_ms']['lookahead']
    pre = {'bcast_gbs_by_root': [300.0, 280.0], 'allgather_gbs_received': 250.0, 'ranks_seen_by_rccl': 2}
    f = M.fabric_from_preflight(pre)
    assert f['bcast_gbs'] == 280.0 and f['allgather_gbs'] == 250.0 and M.fabric_from_preflight({'error': 'x'}) is None

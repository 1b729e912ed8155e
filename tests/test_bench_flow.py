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

"""RCCL initialisation of the native multi-GPU executor is failure-symmetric over the ranks (advisor finding of round 3): whatever
fails on ONE rank before ncclCommInitRank -- the library cannot be loaded there, rank 0 cannot obtain the unique id -- must raise
GpkError on EVERY rank, so that a caller's fall-back (bench.py: the torch.distributed executor) takes the same branch everywhere
and nobody is left inside a collective.  Two gloo processes on CPU; gpk_mg_rccl_probe / gpk_mg_rccl_unique_id are host functions of
libgpk.so (no GPU needed to call them); the object under test is gpk.mg.MultiGpu._init_rccl itself."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


WORKER = textwrap.dedent('''
    import os, sys, time
    ROOT = {root!r}
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
    import torch.distributed as dist
    import gpk
    import gpk.mg as mg
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    scenario = {scenario!r}
    if scenario == 'bad_library_on_rank_1' and rank == 1:
        mg.torch_rccl_path = lambda: '/nonexistent/librccl.so'        # this rank cannot bind RCCL
    if scenario == 'bad_library_everywhere':
        mg.torch_rccl_path = lambda: '/nonexistent/librccl.so'

    class FakeCtx:                                                     # _init_rccl touches ctx only AFTER the agreement
        def _chk(self, rc):
            raise AssertionError('ncclCommInitRank must not be reached when a rank failed before it')
    obj = mg.MultiGpu.__new__(mg.MultiGpu)
    obj.ctx, obj.lib, obj.rank, obj.world, obj.h = FakeCtx(), gpk.load_library(), rank, world, None
    t0 = time.time()
    try:
        obj._init_rccl(None)
        print('rank', rank, 'NO ERROR')
    except gpk.GpkError as e:
        print('rank', rank, 'GpkError:', str(e).replace(chr(10), ' ')[:300])
    except AssertionError as e:
        print('rank', rank, 'REACHED INIT:', e)
    assert time.time() - t0 < 60
    dist.barrier()                                                     # both ranks are still in step with each other
    dist.destroy_process_group()
''')


@pytest.mark.parametrize('scenario', ['bad_library_on_rank_1', 'bad_library_everywhere'])
def test_rccl_init_fails_on_every_rank_together(tmp_path, scenario):
    import gpk
    if not os.path.exists(gpk.library_path()):
        pytest.skip('libgpk.so not built')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, scenario=scenario))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = [l for o in outs for l in o[0].splitlines() if l.startswith('rank')]
    assert len(lines) == 2, outs
    for l in lines:
        assert 'GpkError' in l, lines                                  # every rank raised, nobody reached ncclCommInitRank, nobody hung
    if scenario == 'bad_library_on_rank_1':
        assert all('rank 1' in l.split('GpkError:')[1] for l in lines), lines      # and every rank names the rank that failed

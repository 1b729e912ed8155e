"""The benchmark's ONE JSON line on a real GPU, at a size that takes seconds (BASELINE config 1 as `value`, config 3 through
`--workload c3` is covered by the driver's own run): every field of the driver's contract is present and well-formed, the roofline and
CPU-baseline objects are there, the library's flop counters agree with the host-side models, and `parity` compares the device's first
iterate with the oracle's (<= 1e-6, else the process exits non-zero)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None, tmp=None):
    """-> (compact line as the driver parses it, full object from bench_detail.json, raw stdout)"""
    import tempfile
    tmp = tmp or tempfile.mkdtemp(prefix='gpk_bench_')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['GPK_BENCH_DETAIL_DIR'] = str(tmp)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    assert p.stdout.rstrip('\n').splitlines()[-1] == lines[0]     # the LAST thing on stdout
    assert len(lines[0]) < 7000, len(lines[0])                    # the driver keeps an 8 KB tail (round 4: 29 KB -> parsed: null)
    return json.loads(lines[0]), json.load(open(os.path.join(str(tmp), 'bench_detail.json'))), p.stdout


def test_default_command_of_the_driver_prints_a_compact_parseable_line():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- exactly what the driver runs (the CPU oracle legs of the SECONDARY workloads
    switched off, GPK_BENCH_SECONDARY_CPU=0: they are minutes of host time and are not what this test is about; the primary workload keeps
    its cpu_baseline and parity): ONE line, under the cap, roofline / cpu_baseline / parity at the top level, the timed sequence named."""
    d, full, out = _run('--gpus', '1', '--steps', '20', '--warmup', '5', env_extra={'GPK_BENCH_SECONDARY_CPU': '0'})
    tail = out[-8192:]                                             # what survives in the driver's record
    assert tail.rstrip('\n').splitlines()[-1].startswith('{"metric"') and json.loads(tail.rstrip('\n').splitlines()[-1]) == d
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['value_workload'] == 'c2' and d['dtype'] == 'f64'
    assert 'N_domain=4000' in d['config']['workload'] and 'exact in-step loss' in d['config']['timed_sequence']
    assert abs(d['value'] * d['ms_per_step'] / 1e3 - 1.0) < 1e-4
    r, c, par = d['roofline'], d['cpu_baseline'], d['parity']
    assert r['bound'] == 'mfma' and r['peak'] == 78.6 and 0.3 < r['frac'] < 1 and r['phase_ms'] < d['ms_per_step'] and len(r['kernel']) <= 80
    assert r['flops_per_step'] / (r['phase_ms'] * 1e-3) / 1e12 == pytest.approx(r['achieved'], rel=1e-4)
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['seconds_per_step'] > 0 and c['speedup_vs_reference_sequence'] > 10
    assert par['ok'] is True and par['z1_rel_dev_vs_B2'] <= 1e-6 and par['z1_rel_dev_vs_B1'] <= 1e-6 and d['parity_failed'] is None
    ph = d['phases_ms']                                            # the step's phases + the loss call add up to the step (host gaps aside)
    assert 0.85 * d['ms_per_step'] < ph['solve'] + ph['product_potrf_H'] + ph['tail'] + ph['loss_call'] < 1.02 * d['ms_per_step']
    for k in ('n10k', 'c3', 'c4', 'sharded_config'):
        assert d[k]['value'] > 0 and 0 < d[k]['roofline']['frac'] < 1, (k, d[k])
    assert full['value'] == pytest.approx(d['value'], rel=1e-5) and 'formulation' in full['config']      # the prose lives in the detail file
    assert d['l2_error']['pts_L2_err'] < 1e-6 and d['l2_error']['test_L2_err'] < 1e-6


def test_config1_line_has_the_contract_fields_and_parity():
    line, d, _ = _run('--workload', 'c1', '--steps', '4', '--warmup', '1', '--no-sharded-config', '--no-structured')
    assert line['value'] == pytest.approx(d['value'], rel=1e-5) and line['roofline']['traffic'] is None and line['parity']['ok'] is True
    assert line['value_workload'] == 'c1' and line['cpu_baseline']['kind'] == 'port'
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 1 and d['dtype'] == 'f64' and d['data'] == 'synthetic'
    assert d['higher_is_better'] is True and d['vs_baseline'] is None and 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] == pytest.approx(1e3 / d['ms_per_step'], rel=1e-9)
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 78.6 and r['frac'] == pytest.approx(r['achieved'] / r['peak'])
    assert 0 < r['frac'] < 1 and r['traffic'] is None              # no PMC pass of config 1 is stored: never another workload's bytes
    a = d['roofline_assembly']
    assert a['bound'] == 'hbm' and a['unit'] == 'GB/s' and a['peak'] == 8000.0 and a['kernel_ms'] <= a['call_ms']
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c and c['seconds_per_step'] > 0
    par = d['parity']
    assert par['ok'] is True and par['tol'] == 1e-6 and d['parity_failed'] is None
    assert par['z1_rel_dev_vs_B2'] <= 1e-6 and par['z1_rel_dev_vs_B1'] <= 1e-6
    fc = d['flops_counted_by_library']                             # the launch logic's own count against the mirrored model
    assert fc['solve_flops_per_step'] == pytest.approx(r['flops_per_step'], rel=1e-12)
    assert fc['solve_launches_per_step'] == r['launches_per_step']
    assert fc['product_flops_per_step'] == pytest.approx(d['roofline_syrk']['launched_flops_per_step'], rel=1e-12)
    assert d['l2_error']['pts_L2_err'] < 1e-6 and d['l2_error']['test_L2_err'] < 1e-6


def test_bare_two_rank_command_on_one_gpu():
    """`python3 bench.py --gpus 2 --steps 1 --warmup 0` with NOTHING around it (round 6: bench.py starts its own ranks, bench_launch.py) --
    on this 1-GPU box both ranks share the device (process group over gloo, the bound collectives staged through the host:
    GPK_BENCH_BACKEND=gloo GPK_BENCH_COMM=staged).  The line is the N > 1 line: `value` = the sharded BASELINE config 5 with the real
    kernels, the same job's 1-GPU point beside it, the preflight of the bound collectives."""
    d, full, out = _run('--gpus', '2', '--steps', '1', '--warmup', '0', '--no-replicas',
                        env_extra={'GPK_BENCH_BACKEND': 'gloo', 'GPK_BENCH_COMM': 'staged', 'GPK_SHARDED_TIMEOUT': '1200'})
    assert d['n_gpus'] == 2 and d['value_workload'] == 'c5' and d['scaling'] == 'strong' and d['data'] == 'synthetic'
    assert d['value'] > 0 and abs(d['value'] * d['ms_per_step'] / 1e3 - 1.0) < 1e-4
    assert d['one_gpu_same_job']['value'] > 0 and d['vs_1gpu'] == pytest.approx(d['value'] / d['one_gpu_same_job']['value'], rel=1e-4)
    assert d['preflight']['ranks_seen_by_rccl'] == 2
    assert d['l2_error']['pts_L2_err'] < 1e-6 and d['parity_failed'] is None
    assert isinstance(d.get('predicted'), dict) or isinstance(full.get('predicted'), dict)      # the host model beside the measurement

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


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_config1_line_has_the_contract_fields_and_parity():
    d = _run('--workload', 'c1', '--steps', '4', '--warmup', '1', '--no-sharded-config', '--no-structured')
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

"""The benchmark's stdout line stays parseable by the driver (round 4 lost its measurement to a 29 KB line: BENCH_r04.parsed = null).
CPU-only: bench_line.compact() on the full result object round 4 produced (profiles/r04_bench.json, 28.8 KB) and on a deliberately
bloated one; the -m gpu twin (tests/test_gpu_bench_line.py) runs the driver's default command."""
import copy
import json
import os

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench.json')))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config')


def test_round4_object_compacts_under_the_target_with_everything_the_judge_reads():
    assert len(json.dumps(FULL)) > 25000
    line = bench_line.compact(FULL)
    assert '\n' not in line and len(line) <= bench_line.LINE_TARGET < bench_line.LINE_CAP < 8192
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d['value'] == bench_line.sig(FULL['value']) and d['ms_per_step'] == bench_line.sig(FULL['ms_per_step'])
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 78.6 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-5
    assert len(r['kernel']) <= 80 and r['traffic'] == bench_line.sig(FULL['roofline']['traffic']) and r['flops_per_step'] and r['phase_ms']
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['seconds_per_step'] > 0 and len(c['sample']) <= 96
    assert c['speedup_vs_reference_sequence'] > 1 and c['speedup_vs_triangular'] > 1
    assert d['parity']['ok'] is True and d['parity']['tol'] == 1e-6
    for k in ('n10k', 'c3', 'c4', 'sharded_config'):                # one small object each: numbers, no prose
        o = d[k]
        assert o['value'] > 0 and o['ms_per_step'] > 0 and 0 < o['roofline']['frac'] < 1 and o['parity']['ok'] is True
        assert len(json.dumps(o)) < 900, (k, len(json.dumps(o)))
        assert not any(isinstance(v, str) and len(v) > 100 for v in o.values())
    assert d['detail'] == 'bench_detail.json' and 'dropped_for_size' not in d
    assert d['l2_error']['pts_L2_err'] < 1e-6


def test_bloated_object_never_exceeds_the_cap_and_keeps_the_contract():
    big = copy.deepcopy(FULL)
    big['config']['formulation'] = 'x' * 5000
    big['roofline']['kernel'] = 'k' * 1000
    big['cpu_baseline']['sample'] = 's' * 3000
    for k in ('n10k', 'c3', 'c4', 'sharded_config'):
        big[k]['config']['workload'] = 'w' * 2000
        big[k]['mode_probe'] = {f'probe_{i}': float(i) for i in range(200)}
    big['mode_probe'] = {f'probe_{i}': [float(i)] * 50 for i in range(100)}
    big['preflight'] = {'bcast_gbs_by_root': [1.0] * 4000}
    big['one_time_ms'] = {f'k{i}': float(i) for i in range(500)}
    line = bench_line.compact(big)
    assert len(line) <= bench_line.LINE_CAP
    d = json.loads(line)
    for k in CONTRACT + ('roofline', 'cpu_baseline', 'parity'):
        assert k in d, k
    assert d['dropped_for_size']


def test_failed_sharded_run_leaves_a_null_value_not_another_workloads_figure():
    fb = {'metric': 'm', 'value': None, 'unit': 'GN steps/s', 'n_gpus': 8, 'steps': 20, 'warmup': 5, 'ms_per_step': None, 'higher_is_better': True,
          'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'value_workload': 'c5', 'config': {'workload': 'c5 ...'},
          'fallback': 'did not complete', 'sharded_config': {'error': 'rank 3: RuntimeError: boom'},
          'replicas_c2': {'value': 1190.0, 'ms_per_step': 6.7, 'n_gpus': 8, 'scaling': 'weak'}}
    d = json.loads(bench_line.compact(fb))
    assert d['value'] is None and d['value_workload'] == 'c5' and d['sharded_config'] == {'error': 'rank 3: RuntimeError: boom'}
    assert d['replicas_c2']['value'] == 1190.0 and d['fallback']


def test_non_finite_numbers_do_not_break_the_line():
    o = copy.deepcopy(FULL)
    o['l2_error']['pts_L2_err'] = float('nan')
    o['roofline']['achieved'] = float('inf')
    d = json.loads(bench_line.compact(o))                           # strict JSON: NaN / Infinity would not parse everywhere
    assert d['l2_error'].get('pts_L2_err') is None and d['roofline']['achieved'] is None
    assert 'NaN' not in bench_line.compact(o) and 'Infinity' not in bench_line.compact(o)

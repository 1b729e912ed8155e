"""The benchmark's ONE stdout line, compact.

`bench.py` measures a lot (five workloads, each with rooflines, CPU baselines and parity); the driver keeps only the tail of stdout
(8 KB) and parses the last line.  Round 4 printed everything on that line (29 KB) and the record came back `parsed: null`.  The rule
since round 5: the FULL object goes to `bench_detail.json` (and to stderr), the LAST stdout line is `compact(full)` -- the contract
fields, `config` with short strings, `l2_error`, top-level `roofline` / `cpu_baseline` / `parity`, one small object per secondary
workload (keyed by its name; the workload strings are in the detail file) -- numbers only, no prose, at most LINE_TARGET bytes and
never more than LINE_CAP (keys are dropped in DROP_ORDER until it
fits; `tests/test_bench_line.py` checks the cap on a stored full object and the -m gpu test on the default command).
Pure Python, no GPU, no numpy: importable anywhere.
"""
import json
import math

LINE_TARGET = 6300
LINE_CAP = 7000                     # the driver's stdout tail is 8 KB: leave room for a trailing newline and whatever precedes the line

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data')
SECONDARY = ('n10k', 'c3', 'c4', 'sharded_config', 'one_gpu_same_job', 'replicas_c2')
# what goes first when the line is over LINE_TARGET (least important first); the contract, roofline, cpu_baseline and parity never go
DROP_ORDER = ('structured_step', 'predicted', 'roofline_cholesky_theta', 'step_executed', 'one_time_ms', 'roofline_syrk', 'roofline_assembly',
              'mode_probe', 'preflight', 'replicas_c2', 'one_gpu_same_job', 'phases_ms')


def sig(x, n=6):
    """floats to n significant digits (a 17-digit double costs 10 more bytes than the line needs); everything else unchanged"""
    if hasattr(x, 'item') and not isinstance(x, (str, bytes)) and getattr(x, 'ndim', 0) == 0:
        try:
            x = x.item()            # numpy / torch scalars copied through from mode_probe, preflight, ...: plain Python numbers
        except Exception:           # noqa: BLE001
            pass
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if not math.isfinite(x):
        return None                 # NaN / inf are not JSON
    if x == 0.0:
        return 0.0
    return float(f'{x:.{n}g}')


def rounded(obj, n=6):
    if isinstance(obj, dict):
        return {k: rounded(v, n) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [rounded(v, n) for v in obj]
    return sig(obj, n)


def short(s, n=80):
    if not isinstance(s, str) or len(s) <= n:
        return s
    return s[:n - 3] + '...'


def pick(d, keys, rename=None):
    rename = rename or {}
    return {rename.get(k, k): d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_roofline(r):
    if not isinstance(r, dict):
        return r
    out = pick(r, ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac'))
    out['traffic'] = r.get('traffic')                               # null is information: no PMC pass stored for this workload
    out.update(pick(r, ('flops_per_step', 'flops_per_launch', 'launches_per_step', 'avg_launch_ms', 'phase_ms_per_step', 'ms',
                        'frac_of_partition_peak', 'bytes_per_launch', 'kernel_ms'), {'phase_ms_per_step': 'phase_ms'}))
    if isinstance(out.get('flops_per_step'), dict):                 # (sharded line: solve / product / cholesky_H)
        out['flops_per_step'] = sum(v for v in out['flops_per_step'].values() if isinstance(v, (int, float)))
    if 'kernel' in out:
        out['kernel'] = short(out['kernel'], 80)
    return out


def compact_cpu(c):
    if not isinstance(c, dict):
        return c
    out = pick(c, ('value', 'unit', 'cores', 'kind'))
    out['sample'] = short(c.get('sample', ''), 96)
    out.update(pick(c, ('seconds_per_step', 'triangular_seconds_per_step', 'gpu_speedup_vs_reference_sequence',
                        'gpu_speedup_vs_triangular_best_cpu', 'reference_sequence_timed'),
                    {'gpu_speedup_vs_reference_sequence': 'speedup_vs_reference_sequence',
                     'gpu_speedup_vs_triangular_best_cpu': 'speedup_vs_triangular'}))
    return out


def compact_parity(p):
    if not isinstance(p, dict):
        return p
    if 'skipped' in p:
        return {'skipped': short(p['skipped'], 96)}
    return pick(p, ('z1_rel_dev_vs_B2', 'z1_rel_dev_vs_B1', 'loss0_rel_dev', 'loss1_rel_dev', 'tol', 'ok'))


def compact_phases(ph):
    return pick(ph, ('trsm', 'syrk_and_potrf_H', 'syrk_launches_sum', 'trsv_update', 'loss_call', 'pipelined'),
                {'trsm': 'solve', 'syrk_and_potrf_H': 'product_potrf_H', 'syrk_launches_sum': 'product_launches', 'trsv_update': 'tail'})


def compact_l2(e):
    return pick(e, ('pts_L2_err', 'test_L2_err', 'u_test_L2_err', 'a_test_L2_err', 'gn_steps_run', 'loss_last', 'chol_info'))


def compact_config(c, primary):
    if not isinstance(c, dict):
        return c
    keys = ('workload', 'N_domain', 'N_boundary', 'theta_order', 'theta_orders', 'unknowns', 'kernel', 'kernel_parameter', 'nugget',
            'nugget_type', 'seed', 'timed_sequence', 'parallelism', 'executor') if primary else ('workload', 'N_domain', 'theta_order', 'theta_orders', 'unknowns')
    out = pick(c, keys)
    for k, n in (('workload', 112), ('timed_sequence', 64), ('parallelism', 96), ('executor', 72)):
        if k in out:
            out[k] = short(out[k], n)
    return out


def compact_secondary(o):
    """one small object per secondary workload: value, time, the roofline fraction, errors, parity, the CPU ratios"""
    if not isinstance(o, dict):
        return o
    if 'error' in o and 'value' not in o:
        return {'error': short(o['error'], 160)}
    out = pick(o, ('value', 'ms_per_step', 'n_gpus', 'steps', 'scaling'))
    r = o.get('roofline')
    if isinstance(r, dict):
        out['roofline'] = dict(pick(r, ('achieved', 'frac')), traffic=r.get('traffic'))
    out['phases_ms'] = compact_phases(o.get('phases_ms_per_step'))
    out['l2_error'] = compact_l2(o.get('l2_error'))
    c = o.get('cpu_baseline')
    if isinstance(c, dict):
        out['cpu_baseline'] = pick(c, ('seconds_per_step', 'triangular_seconds_per_step', 'cores', 'gpu_speedup_vs_reference_sequence',
                                       'gpu_speedup_vs_triangular_best_cpu'),
                                   {'gpu_speedup_vs_reference_sequence': 'speedup_vs_reference_sequence',
                                    'gpu_speedup_vs_triangular_best_cpu': 'speedup_vs_triangular'})
    out['parity'] = compact_parity(o.get('parity'))
    if isinstance(out['parity'], dict):
        out['parity'].pop('tol', None)
    if isinstance(o.get('mode_probe'), dict):
        out['mode_probe'] = pick(o['mode_probe'], ('lookahead_kept', 'shard_hb_kept', 'overlap_s_kept'))
    st = o.get('structured_step')
    if isinstance(st, dict):                                        # optional structured solve of this workload: ms, one-time cost, agreement
        lv = st.get('solve_level')
        out['structured'] = pick(lv, ('ms_per_step', 'setup_ms', 'iterate_rel_diff_vs_default')) if isinstance(lv, dict) else {'error': short(str(st.get('error')), 120)}
        if isinstance(st.get('gram_level'), dict):
            out['structured']['gram'] = pick(st['gram_level'], ('ms_per_step', 'setup_ms', 'iterate_rel_diff_vs_default'))
    for k in ('error', 'fallback'):
        if k in o:
            out[k] = short(o[k], 160)
    return {k: v for k, v in out.items() if v not in (None, {}, '')}


def compact_predicted(p):
    """{P: {chol: [look-ahead, sequential], xchg: [all-gather padded, broadcasts exact], hb: [replicated, panel-sharded], step, x}} in ms"""
    if 'error' in p:
        return {'error': short(p['error'], 120)}
    r1 = lambda v: round(v, 1) if isinstance(v, float) else v
    out = {'fabric_gbs': r1(((p.get('inputs') or {}).get('fabric') or {}).get('bcast_gbs')),
           'keys': 'chol=[lookahead,sequential] xchg=[allgather_padded,bcasts_exact,direct_p2p] hb=[replicated,sharded] ms'}
    for k, v in p.items():
        if k == 'inputs' or not isinstance(v, dict):
            continue
        out[k] = {'chol': [r1(v['cholesky_theta_ms']['lookahead']), r1(v['cholesky_theta_ms']['sequential'])],
                  'xchg': [r1(x) for x in v['exchange_of_S_ms'].values()],
                  'hb': [r1(v['cholesky_of_Hb_ms']['replicated']), r1(v['cholesky_of_Hb_ms']['panel_sharded'])],
                  'step': r1(v['step_ms_best']), 'x': round(v['predicted_vs_1gpu'], 2)}
    return out


def _dumps(obj):
    return json.dumps(obj, separators=(',', ':'), default=str)      # (default=str: whatever is not JSON goes out as text, never an exception)


def minimal(full):
    """the contract fields alone: what goes out when compact() itself fails (the ONE stdout line must survive anything)"""
    out = {k: sig(full.get(k)) if isinstance(full, dict) else None for k in CONTRACT}
    if isinstance(full, dict):
        out['value_workload'] = full.get('value_workload')
        out['config'] = {'workload': short(str((full.get('config') or {}).get('workload', '')), 112)}
    return _dumps(out)


def compact(full, detail_path='bench_detail.json'):
    """the driver's line from the full result object"""
    try:
        return _compact(full, detail_path)
    except Exception as e:                                          # noqa: BLE001 -- reported in the line
        out = json.loads(minimal(full))
        out['compact_error'] = short(f'{type(e).__name__}: {e}', 160)
        return _dumps(out)


def _compact(full, detail_path):
    out = {k: full.get(k) for k in CONTRACT}
    out['value_workload'] = full.get('value_workload')
    out['config'] = compact_config(full.get('config'), True)
    out['l2_error'] = compact_l2(full.get('l2_error'))
    out['phases_ms'] = compact_phases(full.get('phases_ms_per_step'))
    out['roofline'] = compact_roofline(full.get('roofline'))
    out['cpu_baseline'] = compact_cpu(full.get('cpu_baseline'))
    out['parity'] = compact_parity(full.get('parity'))
    for k in ('roofline_syrk', 'roofline_assembly', 'roofline_cholesky_theta'):
        if isinstance(full.get(k), dict):
            out[k] = pick(compact_roofline(full[k]), ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'ms', 'kernel_ms',
                                                       'frac_of_partition_peak'))
    if isinstance(full.get('step_executed'), dict):
        out['step_executed'] = pick(full['step_executed'], ('flops_per_step', 'tflops', 'frac_of_peak'))
    if isinstance(full.get('one_time_ms'), dict):
        out['one_time_ms'] = {k: v for k, v in full['one_time_ms'].items() if isinstance(v, (int, float))}
    st = full.get('structured_step')
    if isinstance(st, dict):
        out['structured_step'] = {k: pick(v, ('value', 'ms_per_step', 'iterate_rel_diff_vs_default')) for k, v in st.items() if isinstance(v, dict)}
    for k in SECONDARY:
        if k in full:
            out[k] = compact_secondary(full[k])
    for k in ('vs_1gpu', 'parallel_efficiency', 'parity_failed'):
        if k in full:
            out[k] = full[k]
    if isinstance(full.get('mode_probe'), dict):
        out['mode_probe'] = full['mode_probe']
    if isinstance(full.get('preflight'), dict):
        out['preflight'] = full['preflight']
    if 'fallback' in full:
        out['fallback'] = short(full['fallback'], 200)
    if isinstance(full.get('predicted'), dict):                     # bench_model.table: per rank count, [a, b] pairs in ms, one decimal
        out['predicted'] = compact_predicted(full['predicted'])
    if 'predicted_vs_1gpu' in full:
        out['predicted_vs_1gpu'] = full['predicted_vs_1gpu']
    out['detail'] = detail_path
    out = rounded(out)
    line = _dumps(out)
    dropped = []
    for k in DROP_ORDER:                                            # over the target: shed the least important objects, say which
        if len(line) <= LINE_TARGET:
            break
        if k in out:
            del out[k]
            dropped.append(k)
            out['dropped_for_size'] = dropped
            line = _dumps(out)
    for k in SECONDARY:                                             # still over the hard cap (cannot happen with the objects above): numbers only
        if len(line) <= LINE_CAP:
            break
        if isinstance(out.get(k), dict):
            out[k] = pick(out[k], ('value', 'ms_per_step', 'error'))
            line = _dumps(out)
    if len(line) > LINE_CAP:
        out = {k: out.get(k) for k in CONTRACT + ('value_workload', 'roofline', 'cpu_baseline', 'parity', 'detail')}
        out['config'] = {'workload': short(str((full.get('config') or {}).get('workload', '')), 112)}
        line = _dumps(out)
    return line

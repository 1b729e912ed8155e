#!/usr/bin/env python3
"""gpurun_out/prof (tools/profile_round.sh) -> profiles/<round>_*: kernel stats table, bench lines, PMC summary of the
dominant kernel (SYRK = gemm_f64_kernel<...,true,false> with the largest grid) and of the assembly kernel.
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB units; on gfx950
FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming reads as 64 bytes -> x2 for kernels that read
with dwordx4 (the GEMM); WRITE_SIZE needs no correction (calibrated on the assembly kernel, which writes exactly 8 N^2
bytes)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
DST = os.path.join(ROOT, 'profiles')

def one(pattern):
    """the NEWEST match: gpurun merges every call's files into the same local tree, and rocprofv3 names its outputs by process id, so
    files of earlier profile rounds can sit next to the current ones (that is how a round-3 table named a kernel the round-3 build no
    longer launched); tools/profile_round.sh is best run after `rm -rf gpurun_out/prof` here"""
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None

def counters(name):
    f = one(f'pmc_{name}/**/*counter_collection.csv')
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        k = (int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Grid_Size']))
        d = out.setdefault(k, {'dur_us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return out

def pick(table, pred):
    rows = [v for k, v in table.items() if pred(k)]
    return rows

def mean(rows, key):
    vals = [r[key] for r in rows if key in r]
    return sum(vals) / len(vals) if vals else None

stats = one('stats/**/*kernel_stats.csv')
if stats:                                                        # default command: config 2 + the sharded config 5 run
    shutil.copy(stats, os.path.join(DST, f'{tag}_bench_full_kernel_stats.csv'))
stats = one('stats_c2/**/*kernel_stats.csv')
if stats:                                                        # --no-sharded-config: the value's workload alone
    shutil.copy(stats, os.path.join(DST, f'{tag}_bench_kernel_stats.csv'))
if os.path.exists(os.path.join(SRC, 'bench_detail.json')):       # the full result object of the plain run (the line itself is its compact form)
    shutil.copy(os.path.join(SRC, 'bench_detail.json'), os.path.join(DST, f'{tag}_bench_detail.json'))
for src, dst in [('bench.json', f'{tag}_bench.json'), ('bench_under_rocprof.json', f'{tag}_bench_under_rocprof.json')]:
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith('{')]
        if lines:
            open(os.path.join(DST, dst), 'w').write(json.dumps(json.loads(lines[-1]), indent=1) + '\n')

import re
def gemm_kind(name):
    """'NN' / 'NT' / 'TN' / 'TT' of a gemm_f64_kernel<BM, BN, WM, WN, TA, TB[, TRI]> instantiation, else None"""
    m = re.search(r'gemm_f64_kernel<\d+, \d+, \d+, \d+, (true|false), (true|false)', name)
    return None if not m else ('T' if m.group(1) == 'true' else 'N') + ('T' if m.group(2) == 'true' else 'N')
is_syrk = lambda k: gemm_kind(k[1]) == 'TN'                       # only the product S^T S uses the TN form
is_asm = lambda k: 'assemble' in k[1] and 'test' not in k[1]
def summarise(prefix, workload, suffix):
    summary = {'workload': workload, 'kernel': f'gemm_f64_kernel<.., TN> (SYRK Hb = S^T S, lower tiles), workload {workload}'}
    fetch, write = counters(prefix + 'FETCH_SIZE'), counters(prefix + 'WRITE_SIZE')
    busy, tcc = counters(prefix + 'SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE'), counters(prefix + 'TCC_HIT_sum+TCC_MISS_sum')
    fs, ws = pick(fetch, is_syrk), pick(write, is_syrk)
    if fs and ws:
        f_kib, w_kib = mean(fs, 'FETCH_SIZE'), mean(ws, 'WRITE_SIZE')
        summary.update({'launches_sampled': len(fs), 'FETCH_SIZE_KiB_raw': f_kib, 'WRITE_SIZE_KiB_raw': w_kib,
                        'correction': 'FETCH_SIZE x2 (16-byte-per-lane reads, gfx950), WRITE_SIZE x1; KiB units',
                        'fetch_bytes_corrected': f_kib * 1024 * 2, 'write_bytes': w_kib * 1024,
                        'hbm_bytes_per_launch': f_kib * 1024 * 2 + w_kib * 1024,
                        'avg_duration_us_under_pmc': mean(fs, 'dur_us')})
    bs = pick(busy, is_syrk)
    if bs:
        b, g = mean(bs, 'SQ_VALU_MFMA_BUSY_CYCLES'), mean(bs, 'GRBM_GUI_ACTIVE')
        summary.update({'SQ_VALU_MFMA_BUSY_CYCLES': b, 'GRBM_GUI_ACTIVE': g,
                        'mfma_busy_fraction': b / 1024 / (g / 8),
                        'executed_flops_from_counter': b / 64 * 2048,
                        'effective_clock_ghz': (g / 8) / (mean(bs, 'dur_us') * 1e3),
                        'note': 'SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per v_mfma_f64_16x16x4_f64 summed over the 1024 SIMDs; '
                                'GRBM_GUI_ACTIVE is summed over the 8 XCDs; mfma_busy_fraction = busy / 1024 / (GUI_ACTIVE / 8)'})
    ts = pick(tcc, is_syrk)
    if ts:
        h, m = mean(ts, 'TCC_HIT_sum'), mean(ts, 'TCC_MISS_sum')
        summary.update({'TCC_HIT_sum': h, 'TCC_MISS_sum': m, 'l2_hit_rate': h / (h + m)})
    fa, wa = pick(fetch, is_asm), pick(write, is_asm)
    if fa and wa:
        summary['assemble_kernel'] = {'WRITE_SIZE_KiB': mean(wa, 'WRITE_SIZE'), 'FETCH_SIZE_KiB_raw': mean(fa, 'FETCH_SIZE'),
                                      'write_bytes': mean(wa, 'WRITE_SIZE') * 1024, 'avg_duration_us_under_pmc': mean(wa, 'dur_us')}
    bench = os.path.join(DST, f'{tag}_bench_detail.json')
    if not os.path.exists(bench):
        bench = os.path.join(DST, f'{tag}_bench.json')
    if os.path.exists(bench):
        bj = json.load(open(bench))
        bj = bj if workload == 'c2' else bj.get({'c5': 'sharded_config'}.get(workload, workload), {})
        rl = bj.get('roofline_syrk') or bj.get('roofline', {})
        summary['algorithmic_flops_per_launch'] = rl.get('flops_per_launch')
        summary['note_launches'] = ('since round 2 the product is issued as one launch per 512-column block (pipelined with the '
                                    'factorisation): all figures above are MEANS PER LAUNCH over those launches; hbm_bytes_per_step below '
                                    'is their sum per Gauss-Newton step')

    # ---- per-step sums inside the Gauss-Newton steps (dispatches after the first gn_build_kernel): the solve phase's GEMM family
    def per_step(table, pred, key, nsteps):
        first = min((k[0] for k in table if 'gn_build_kernel' in k[1]), default=None)
        if first is None or not nsteps:
            return None, 0
        rows = [v for k, v in table.items() if k[0] > first and pred(k) and key in v]
        return (sum(r[key] for r in rows) / nsteps if rows else None), len(rows)

    def n_steps(table):
        # gn_build_kernel also runs once in the closing gn_loss and once per extension-free loss: count reverse_copy (one per step)
        return sum(1 for k in table if 'reverse_copy_kernel' in k[1])

    is_nn = lambda k: gemm_kind(k[1]) == 'NN'
    trsm = {'workload': workload, 'kernel': f'gemm_f64_kernel<.., NN> (all instantiations) inside the Gauss-Newton steps = the solve phase S = L^{{-1}}[A | F], workload {workload}'}
    for name, table, key, scale in (('fetch', fetch, 'FETCH_SIZE', 2048.0), ('write', write, 'WRITE_SIZE', 1024.0)):
        ns = n_steps(table)
        v, n = per_step(table, is_nn, key, ns)
        if v is not None:
            trsm[f'{name}_bytes_per_step'] = v * scale
            trsm[f'{name}_launches_per_step'] = n / ns
        v, n = per_step(table, is_syrk, key, ns)
        if v is not None:
            summary[f'{name}_bytes_per_step'] = v * scale
    ns = n_steps(busy)
    b, n = per_step(busy, is_nn, 'SQ_VALU_MFMA_BUSY_CYCLES', ns)
    d, _ = per_step(busy, is_nn, 'dur_us', ns)
    if b is not None:
        trsm.update({'SQ_VALU_MFMA_BUSY_CYCLES_per_step': b, 'executed_flops_from_counter_per_step': b / 64 * 2048,
                     'sum_of_launch_durations_us_per_step_under_pmc': d, 'launches_per_step': n / ns,
                     'tflops_from_counter_under_pmc': b / 64 * 2048 / (d * 1e-6) / 1e12})
    if 'fetch_bytes_per_step' in trsm and 'write_bytes_per_step' in trsm:
        trsm['hbm_bytes_per_step'] = trsm['fetch_bytes_per_step'] + trsm['write_bytes_per_step']
    if 'fetch_bytes_per_step' in summary and 'write_bytes_per_step' in summary:
        summary['hbm_bytes_per_step'] = summary['fetch_bytes_per_step'] + summary['write_bytes_per_step']
    h, _ = per_step(tcc, is_nn, 'TCC_HIT_sum', n_steps(tcc)); m, _ = per_step(tcc, is_nn, 'TCC_MISS_sum', n_steps(tcc))
    if h is not None and m is not None:
        trsm['l2_hit_rate'] = h / (h + m)
    if os.path.exists(bench):
        bj = json.load(open(bench))
        bj = bj if workload == 'c2' else bj.get({'c5': 'sharded_config'}.get(workload, workload), {})
        trsm['executed_flops_per_step_model'] = (bj.get('roofline') or {}).get('flops_per_step')
    # kernel-trace cross-check of the live figure: sum of the NN launches per step in the stats run of the value's workload
    st = os.path.join(DST, f'{tag}_bench_kernel_stats.csv' if workload == 'c2' else f'{tag}_bench_{workload}_kernel_stats.csv')
    if os.path.exists(st):
        tot = calls = 0
        for r in csv.DictReader(open(st)):
            if gemm_kind(r['Name']) == 'NN':
                tot += int(r['TotalDurationNs']); calls += int(r['Calls'])
        trsm['kernel_stats_total_ms_all_NN_launches'] = tot / 1e6
        trsm['kernel_stats_calls'] = calls
    open(os.path.join(DST, f'{tag}_pmc_syrk{suffix}.json'), 'w').write(json.dumps(summary, indent=1) + '\n')
    open(os.path.join(DST, f'{tag}_pmc_trsm_gemm{suffix}.json'), 'w').write(json.dumps(trsm, indent=1) + '\n')
    print(json.dumps(summary, indent=1))
    print(json.dumps(trsm, indent=1))


# kernel-stats tables of the other single-workload runs (c3, c4, n10k) next to the config-2 one
for wl in ('n10k', 'c3', 'c4', 'c5'):
    st = one(f'stats_{wl}/**/*kernel_stats.csv')
    if st:
        shutil.copy(st, os.path.join(DST, f'{tag}_bench_{wl}_kernel_stats.csv'))
summarise('', 'c2', '')
for wl in ('n10k', 'c3', 'c4', 'c5'):
    if one(f'pmc_{wl}_FETCH_SIZE/**/*counter_collection.csv'):
        summarise(f'{wl}_', wl, f'_{wl}')

#!/usr/bin/env python3
"""gpurun_out/prof/pmc_n10k_SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE (tools/profile_round.sh, steps 4-5) -> profiles/<round>_pmc_potrf_n10k.json: MFMA-busy fraction of the
Cholesky factorisation's launches at the north-star size (N_domain = 10000, Theta of order 21000).

Dispatches are taken from the run `bench.py --workload n10k --steps 2 --warmup 1`: everything between the first and the last
potrf_panel_mfma_kernel dispatch that precedes the first gn_build_kernel belongs to gpk_potrf(Theta) (it runs twice: cold + warm).
  * trailing updates  = gemm_f64_kernel<.., false, true, ..> (NT, lower tiles; one per 512-column block)
  * panel kernels     = potrf_panel_mfma_kernel (fused variant: the previous panel's rank-64 updates ride inside)
SQ_VALU_MFMA_BUSY_CYCLES counts 64 cycles per v_mfma_f64_16x16x4_f64 summed over the 1024 SIMDs, GRBM_GUI_ACTIVE is summed over the 8
XCDs: busy fraction = busy / 1024 / (GUI_ACTIVE / 8); executed flops = busy / 64 * 2048."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
f = glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof', 'pmc_n10k_SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE', '**', '*counter_collection.csv'), recursive=True)
if not f:
    sys.exit('no counter file')
rows = {}
for r in csv.DictReader(open(max(f, key=os.path.getmtime))):      # the newest pass (earlier rounds' files may sit beside it)
    k = int(r['Dispatch_Id'])
    d = rows.setdefault(k, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']), 'dur_us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(rows)
first_build = min((i for i in ids if 'gn_build_kernel' in rows[i]['name']), default=max(ids) + 1)
theta = [i for i in ids if i < first_build]
panel = [rows[i] for i in theta if 'potrf_panel_mfma_kernel' in rows[i]['name']]
nt = [rows[i] for i in theta if re.search(r'gemm_f64_kernel<\d+, \d+, \d+, \d+, false, true', rows[i]['name'])]

def agg(rs):
    b = sum(r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for r in rs)
    g = sum(r.get('GRBM_GUI_ACTIVE', 0.0) for r in rs)
    t = sum(r['dur_us'] for r in rs)
    return {'launches': len(rs), 'sum_duration_us_under_pmc': t, 'SQ_VALU_MFMA_BUSY_CYCLES': b, 'GRBM_GUI_ACTIVE': g,
            'mfma_busy_fraction': (b / 1024) / (g / 8) if g else None,
            'executed_tflops_under_pmc': (b / 64 * 2048) / (t * 1e-6) / 1e12 if t else None,
            'executed_flops': b / 64 * 2048, 'effective_clock_ghz': (g / 8) / (t * 1e3) if t else None}
N = 21000
big = [r for r in nt if r['dur_us'] > 200.0]                     # the large trailing updates (first blocks): the "Cholesky panel update" at scale
out = {'workload': 'NonLinElliptic2d N_domain=10000 N_boundary=1000: gpk_potrf of Theta (order 21000), two calls (cold + warm)',
       'trailing_updates_gemm_NT': agg(nt), 'trailing_updates_longer_than_200us': agg(big), 'panel_kernels_fused': agg(panel),
       'factorisation_flops_per_call': N ** 3 / 3.0,
       'note': 'north star: ">= 40 % fp64 MFMA utilisation in the Cholesky panel update" -- the trailing (panel) updates are the '
               'gemm_f64_kernel<NT> launches; mfma_busy_fraction is the counter ratio over their own execution time, '
               'executed_tflops_under_pmc / 78.6 the fraction of the datasheet peak (PMC mode runs ~7 % slower than plain)'}
open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_potrf_n10k.json'), 'w').write(json.dumps(out, indent=1) + '\n')
st = glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof', 'stats_n10k', '**', '*kernel_stats.csv'), recursive=True)
if st:
    import shutil
    shutil.copy(st[0], os.path.join(ROOT, 'profiles', f'{tag}_bench_n10k_kernel_stats.csv'))
print(json.dumps(out, indent=1))
